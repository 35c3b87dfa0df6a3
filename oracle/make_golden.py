#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE's own code from /root/reference.

Runs only in the build container (the reference is not on the GPU box).  The third-party packages
the reference imports at module load (gym, ase, torch_scatter, schnetpack, quadpy, cormorant) are
absent; they are replaced by import-time stubs.  Only two of them carry arithmetic that the pinned
functions touch, and those get functional stand-ins built from independent libraries:
  * cormorant.cg_lib.SphericalHarmonics  -> closed-form Y_lm from scipy.special.sph_harm_y
    (the reference pins this convention itself, tests/agents/covariant/test_sphs.py)
  * quadpy.u3._lebedev.lebedev_071       -> scipy.integrate.lebedev_rule(71), weights / 4 pi
Everything written to the fixtures is OUTPUT of reference functions:
  molgym.ppo.compute_loss / get_batch_generator, molgym.buffer.DynamicPPOBuffer,
  molgym.tools.util.discount_cumsum / compute_gradient_norm, molgym.modules.MLP / to_one_hot,
  molgym.agents.covariant.gmm.GaussianMixtureModel, so3_tools.*, spherical_dists.*
Usage: python oracle/make_golden.py
"""
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
np.float = float  # removed numpy aliases the reference still uses
np.product = np.prod


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


class SO3Vec(list):
    @property
    def ells(self):
        return [(p.shape[-2] - 1) // 2 for p in self]


class SphHarm:
    """stand-in for cormorant.cg_lib.SphericalHarmonics (sh_norm 'qm', normalize=True, conj=False)."""

    def __init__(self, maxl, sh_norm='qm', **kw):
        self.maxl, self.sh_norm = maxl, sh_norm

    def forward(self, pos):
        from scipy.special import sph_harm_y
        shape = pos.shape[:-1]
        p = pos.detach().double().reshape(-1, 3)
        n = p.norm(dim=-1, keepdim=True)
        p = torch.where(n > 0, p / n, torch.zeros_like(p)).numpy()
        theta = np.arccos(np.clip(p[:, 2], -1, 1))
        phi = np.arctan2(p[:, 1], p[:, 0])
        parts = []
        for l in range(self.maxl + 1):
            y = np.stack([sph_harm_y(l, m, theta, phi) for m in range(-l, l + 1)], axis=-1)
            t = torch.tensor(np.stack([y.real, y.imag], axis=-1), dtype=pos.dtype)
            parts.append(t.reshape(shape + (1, 2 * l + 1, 2)))
        return SO3Vec(parts)


def install_stubs():
    _stub('gym', Env=object, spaces=_stub('gym.spaces', Tuple=object, Discrete=object, Box=object))
    ase = _stub('ase', Atom=object, Atoms=object)
    ase.data = _stub('ase.data', atomic_numbers={}, chemical_symbols=[])
    ase.io = _stub('ase.io')
    ase.formula = _stub('ase.formula', Formula=object)
    _stub('torch_scatter', composite=_stub('torch_scatter.composite'))
    _stub('schnetpack')
    _stub('scipy.spatial.qhull', QhullError=Exception)
    cg_lib = _stub('cormorant.cg_lib', CGModule=torch.nn.Module, SphericalHarmonics=SphHarm,
                   SphericalHarmonicsRel=object, CGProduct=object, CGDict=object)
    so3_lib = _stub('cormorant.so3_lib', SO3Vec=SO3Vec, SO3Tau=list, rotations=None)
    nn_ = _stub('cormorant.nn', NoLayer=object, RadialFilters=object, CatMixReps=object, InputLinear=object)
    models = _stub('cormorant.models')
    models.cormorant_cg = _stub('cormorant.models.cormorant_cg', CormorantCG=object)
    models.cormorant_qm9 = _stub('cormorant.models.cormorant_qm9', expand_var_list=None)
    _stub('cormorant', cg_lib=cg_lib, so3_lib=so3_lib, nn=nn_, models=models)
    from scipy.integrate import lebedev_rule

    class _Grid:
        def __init__(self):
            x, w = lebedev_rule(71)
            self.points, self.weights = x, w / (4 * math.pi)

    leb = _stub('quadpy.u3._lebedev', lebedev_071=_Grid)
    u3 = _stub('quadpy.u3', _lebedev=leb)
    _stub('quadpy', u3=u3)
    for mod in ('scine_sparrow', 'scine_utilities'):
        _stub(mod)


def main():
    install_stubs()
    sys.path.insert(0, '/root/reference')
    from molgym import ppo
    from molgym.agents.covariant import so3_tools, spherical_dists
    from molgym.agents.covariant.gmm import GaussianMixtureModel
    from molgym.buffer import DynamicPPOBuffer
    from molgym.modules import MLP, to_one_hot
    from molgym.tools import util
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(1234)

    # ---- G1 compute_loss -----------------------------------------------------------------------
    class FakeAC:
        def __init__(self, logp, ent, v):
            self.t = [torch.tensor(x, dtype=torch.float32, requires_grad=True) for x in (logp, ent, v)]

        def step(self, obs, act):
            return {'logp': self.t[0], 'ent': self.t[1], 'v': self.t[2]}

    g1 = {}
    for B in (1, 7, 140):
        logp_old = rng.normal(-5, 1, B)
        logp = (logp_old + rng.normal(0, 0.4, B)).astype(np.float32)  # ratios inside and outside the clip band
        ent, v = rng.uniform(0, 3, B).astype(np.float32), rng.normal(0, 1, B).astype(np.float32)
        adv, ret = rng.normal(0, 1, B), rng.normal(0, 0.3, B)
        ac = FakeAC(logp, ent, v)
        loss, info = ppo.compute_loss(ac, dict(obs=None, act=None, logp=logp_old, adv=adv, ret=ret), clip_ratio=0.2,
                                      vf_coef=0.5, entropy_coef=0.01)
        loss.backward()
        assert loss.dtype == torch.float64
        keys = ['policy_loss', 'entropy_loss', 'vf_loss', 'total_loss', 'approx_kl', 'clip_fraction']
        g1.update({f'B{B}_logp': logp, f'B{B}_ent': ent, f'B{B}_v': v, f'B{B}_old_logp': logp_old, f'B{B}_adv': adv,
                   f'B{B}_ret': ret, f'B{B}_stats': np.array([info[k] for k in keys], dtype=np.float64),
                   f'B{B}_grads': np.stack([t.grad.numpy() for t in ac.t])})
    np.savez(os.path.join(OUT, 'g1_compute_loss.npz'), **g1)

    # ---- G2 GAE / advantage normalisation -------------------------------------------------------
    g2 = {}
    case = 0
    for gamma in (1.0, 0.99):
        buf = DynamicPPOBuffer(gamma=gamma, lam=0.97)
        offs, lasts = [0], []
        for T in (1, 3, 7, 4):
            for t in range(T):
                buf.store(obs=None, act=np.zeros(6), reward=float(rng.normal()), next_obs=None, terminal=t == T - 1,
                          value=float(rng.normal()), logp=float(rng.normal()))
            last = 0.0 if T != 4 else float(rng.normal())  # last path is cut mid-episode
            buf.finish_path(last)
            offs.append(buf.current_index)
            lasts.append(last)
        data = buf.get_data()
        g2.update({f'c{case}_gamma': gamma, f'c{case}_lam': 0.97, f'c{case}_off': np.array(offs, dtype=np.int32),
                   f'c{case}_rew': np.array(buf.rew_buf), f'c{case}_val': np.array(buf.val_buf),
                   f'c{case}_last': np.array(lasts), f'c{case}_adv': np.array(buf.adv_buf),
                   f'c{case}_ret': np.array(buf.ret_buf), f'c{case}_adv_norm': data['adv']})
        case += 1
    g2['dc_x'] = np.array([1.0, 1.0, 1.0])
    g2['dc_y'] = util.discount_cumsum(np.array([1.0, 1.0, 1.0]), 0.5)
    np.savez(os.path.join(OUT, 'g2_gae.npz'), **g2)

    # ---- G3 mini-batch index generator ----------------------------------------------------------
    g3 = {}
    for i, (T, mb, seed) in enumerate(((140, 140, 0), (140, 64, 1), (10, 3, 2))):
        np.random.seed(seed)
        batches = list(ppo.get_batch_generator(np.arange(T), mb))
        g3[f'c{i}_T'], g3[f'c{i}_mb'], g3[f'c{i}_seed'] = T, mb, seed
        g3[f'c{i}_n'] = len(batches)
        for j, b in enumerate(batches):
            g3[f'c{i}_b{j}'] = b
    np.savez(os.path.join(OUT, 'g3_batches.npz'), **g3)

    # ---- G4 MLP, G9 one-hot / grad norm ----------------------------------------------------------
    torch.manual_seed(7)
    mlp = MLP(input_dim=12, output_dims=(16, 5))
    x = torch.randn(9, 12, requires_grad=True)
    y = mlp(x)
    (y * torch.linspace(-1, 1, 45).view(9, 5)).sum().backward()
    g4 = {f'sd_{k}': v.numpy() for k, v in mlp.state_dict().items()}
    g4.update(x=x.detach().numpy(), y=y.detach().numpy(), dx=x.grad.numpy(),
              **{f'grad_{k}': p.grad.numpy() for k, p in mlp.named_parameters()})
    g4['grad_norm'] = util.compute_gradient_norm(mlp.parameters())
    idx = torch.tensor([[0], [3], [2]])
    g4['oh_idx'], g4['oh'] = idx.numpy(), to_one_hot(idx, num_classes=4).numpy()
    np.savez(os.path.join(OUT, 'g4_mlp.npz'), **g4)

    # ---- G5 GMM ------------------------------------------------------------------------------
    lp = torch.randn(11, 3, requires_grad=True)
    means = (torch.rand(11, 3) * 1.0 + 0.8).requires_grad_(True)
    log_stds = torch.log(torch.tensor([0.1, 0.2, 0.05])).requires_grad_(True)
    xs = torch.rand(11) * 1.2 + 0.7
    gmm = GaussianMixtureModel(log_probs=lp, means=means, stds=torch.exp(log_stds).clamp(1e-6))
    out = gmm.log_prob(xs)
    out.sum().backward()
    np.savez(os.path.join(OUT, 'g5_gmm.npz'), log_probs=lp.detach().numpy(), means=means.detach().numpy(),
             log_stds=log_stds.detach().numpy(), x=xs.numpy(), logp=out.detach().numpy(), d_log_probs=lp.grad.numpy(),
             d_means=means.grad.numpy(), d_log_stds=log_stds.grad.numpy())

    # ---- G6 so3 tools ----------------------------------------------------------------------------
    def rand_vec(batch, tau):
        return SO3Vec([torch.randn(*batch, tau, 2 * l + 1, 2) for l in range(5)])

    g6 = {}
    scal = so3_tools.AtomicScalars(maxl=4)
    for tau in (4, 12):
        v = rand_vec((3, ), tau)
        for l, p in enumerate(v):
            g6[f't{tau}_in_{l}'] = p.numpy()
        g6[f't{tau}_scalars'] = scal(v).numpy()
        n = so3_tools.normalize_alms(v)
        for l, p in enumerate(n):
            g6[f't{tau}_norm_{l}'] = p.numpy()
        g6[f't{tau}_k'] = so3_tools.get_normalization_constant(v).numpy()
    a, y = rand_vec((3, ), 4), SO3Vec([torch.randn(3, 1, 2 * l + 1, 2) for l in range(5)])
    for l in range(5):
        g6[f'sp_a_{l}'], g6[f'sp_y_{l}'] = a[l].numpy(), y[l].numpy()
    g6['sp_out'] = so3_tools.sum_product_alms_ylms(a, y).numpy()
    v5 = rand_vec((3, 5), 12)
    focus = torch.zeros(3, 5)
    focus[0, 1] = focus[1, 4] = focus[2, 0] = 1
    sel = so3_tools.select_atomic_covariats(v5, focus)
    indices = torch.tensor([[4, 5, 6, 7], [0, 1, 2, 3], [8, 9, 10, 11]])
    tau_sel = so3_tools.select_taus(sel, indices)
    for l in range(5):
        g6[f'sel_in_{l}'], g6[f'sel_cov_{l}'], g6[f'sel_tau_{l}'] = v5[l].numpy(), sel[l].numpy(), tau_sel[l].numpy()
    g6['sel_focus'], g6['sel_indices'] = focus.numpy(), indices.numpy()
    g6['fib16'] = so3_tools.generate_fibonacci_grid(16)
    np.savez(os.path.join(OUT, 'g6_so3_tools.npz'), **g6)

    # ---- G7 spherical distributions ---------------------------------------------------------------
    g7 = {}
    sphs = SphHarm(maxl=4, sh_norm='qm')
    a = rand_vec((6, ), 4)
    dirs = torch.randn(6, 3)
    for l in range(5):
        g7[f'a_{l}'] = a[l].numpy()
    g7['dirs'] = dirs.numpy()
    for beta in (-10.0, 100.0):
        d = spherical_dists.ExpSO3Distribution(a_lms=a, sphs=sphs, beta=beta, dtype=torch.float32)
        tag = 'm10' if beta < 0 else 'p100'
        g7[f'exp_{tag}_logz'], g7[f'exp_{tag}_logp'] = d.log_z.numpy(), d.log_prob(dirs).numpy()
    empty = torch.tensor([False, True, False, False, True, False])
    d0 = spherical_dists.SO3Distribution(a_lms=a, sphs=sphs, empty=empty, dtype=torch.float32)
    d1 = spherical_dists.SO3Distribution(a_lms=a, sphs=sphs, empty=None, dtype=torch.float32)
    g7['so3_empty'], g7['so3_logp_empty'], g7['so3_logp'] = empty.numpy(), d0.log_prob(dirs).numpy(), d1.log_prob(dirs).numpy()
    from scipy.integrate import lebedev_rule
    pts, w = lebedev_rule(71)
    g7['leb_points'], g7['leb_weights'] = pts.T.astype(np.float64), (w / (4 * math.pi)).astype(np.float64)
    np.savez_compressed(os.path.join(OUT, 'g7_spherical.npz'), **g7)
    # literal known answers of the reference's own tests
    np.savez(os.path.join(OUT, 'known_answers.npz'),
             sph_l1_pos=so3_tools.spherical_to_cartesian(np.array([np.pi / 2, 0.0])),
             sph_l1=np.array([[0.345494, 0], [0, 0], [-0.345494, 0]]),  # test_sphs.py:28-32
             sph_l2_pos=so3_tools.spherical_to_cartesian(np.array([np.pi / 3, np.pi / 4])),
             sph_l2=np.array([[0, -0.289706], [0.236544, -0.236544], [-0.0788479, 0], [-0.236544, -0.236544],
                              [0, 0.289706]]),  # test_sphs.py:46-53
             complex_prod=so3_tools.complex_product(torch.tensor([2., -1.]), torch.tensor([3., -2.])).numpy())
    # ---- G8 z-matrix placement (internal/zmat.py) ------------------------------------------------
    from molgym.agents.internal import zmat
    g8 = {}
    for n in (0, 1, 2, 3, 5):
        pos = [rng.normal(size=3) * 1.5 for _ in range(n)]
        focus = int(rng.integers(0, max(n, 1)))
        d, ang, dih = float(rng.uniform(0.9, 1.8)), float(rng.uniform(0.3, 2.8)), float(rng.uniform(-3.0, 3.0))
        g8[f'n{n}_pos'] = np.array(pos).reshape(n, 3)
        g8[f'n{n}_args'] = np.array([focus, d, ang, dih])
        g8[f'n{n}_out'] = zmat.position_atom_helper(positions=pos, focus=focus, distance=d, angle=ang, dihedral=dih)
    p = [rng.normal(size=3) for _ in range(4)]
    g8['geo_pts'] = np.array(p)
    g8['geo'] = np.array([zmat.get_distance(p[0], p[1]), zmat.get_angle(p[0], p[1], p[2]),
                          zmat.get_dihedral(p[0], p[1], p[2], p[3])])
    np.savez(os.path.join(OUT, 'g8_zmat.npz'), **g8)
    print('golden fixtures written to', OUT)


if __name__ == '__main__':
    main()
