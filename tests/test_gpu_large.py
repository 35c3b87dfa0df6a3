"""Parity at BASELINE's full sizes, where the float64 oracle is out of reach: size-independent properties of the
path (SO(3) invariance, independence of the samples of a mini-batch, additivity of the gradient over shards) and
agreement of the kernel families that only large shapes select (MFMA forms vs the lane-per-row VALU forms, the
16-byte dW form, the LDS-staged row GEMM, the two-kernel list build) on the same inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from molgym_amd.agents.covariant import CovariantAC
from molgym_amd.spaces import ActionSpace, ObservationSpace
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _agent(cfg, seed=0):
    torch.manual_seed(seed)
    return CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']),
                       bag_scale=cfg['bag_scale'], beta=cfg['beta'], device=torch.device('cuda:0'), **MODEL_DEFAULTS)


def _rotation(seed):
    q, r = np.linalg.qr(np.random.default_rng(seed).normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _rotate(data, R):
    obs = []
    for canvas, bag in data['obs']:
        obs.append((tuple((label, tuple(R @ np.asarray(xyz, dtype=np.float64))) for label, xyz in canvas), bag))
    act = np.array(data['act'], dtype=np.float64)
    act[:, 3:6] = act[:, 3:6] @ R.T
    return obs, act


@pytest.mark.parametrize('name', ['cfg3', 'cfg5'])
def test_rotation_invariance_at_full_size(built_lib, name):
    """a global rotation of canvas and orientation leaves logp, entropy and value unchanged (test_agent.py:43-123
    checks the same property of the reference at toy size)"""
    cfg = CONFIGS[name]
    ac = _agent(cfg)
    data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=5)
    with torch.no_grad():
        a = ac.forward_batch(ac.prepare_batch(data['obs'], data['act'])).clone()
        obs_r, act_r = _rotate(data, _rotation(1))
        b = ac.forward_batch(ac.prepare_batch(obs_r, act_r)).clone()
    assert torch.isfinite(a).all()
    scale = a.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
    assert ((a - b).abs() / scale).max().item() < 2e-4  # f32 evaluation of a 3-level CG network at N = 12 / 40


@pytest.mark.parametrize('name', ['cfg3', 'cfg5'])
def test_samples_are_independent_at_full_size(built_lib, name):
    """the outputs of a sample do not depend on the rest of the mini-batch (ragged compaction, tiling, atomics)"""
    cfg = CONFIGS[name]
    ac = _agent(cfg)
    data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=6)
    with torch.no_grad():
        full = ac.forward_batch(ac.prepare_batch(data['obs'], data['act'])).clone()
        lo, hi = cfg['batch'] // 3, cfg['batch'] // 3 + 37
        part = ac.forward_batch(ac.prepare_batch(data['obs'][lo:hi], data['act'][lo:hi])).clone()
    # forward sums run in a fixed order per sample, so the only difference is the tiling of the MFMA row blocks
    assert (full[:, lo:hi] - part).abs().max().item() < 1e-5 * max(1.0, full.abs().max().item())


@pytest.mark.parametrize('name', ['cfg3', 'cfg5'])
def test_gradient_is_additive_over_shards_at_full_size(built_lib, name):
    """data-parallel contract at full size: the gradient of a mini-batch is the sum of its shards'
    gradients with loss scale B_shard / B (cfg5: 2048 samples on canvases of 40 -- the > 512-sample heads backward, the
    16-byte dW form, the shared DotMatrix block and slice offsets near their 32-bit limit on one side, their smaller-shape
    siblings on the other)"""
    cfg = CONFIGS[name]
    ac = _agent(cfg)
    B = cfg['batch']
    data = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=7)
    cut = B // 2 + 17
    ac.theta.grad = torch.zeros_like(ac.theta)
    ac.ppo_minibatch(ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret']), 0.2, 0.5, 0.01)
    torch.cuda.synchronize()
    whole = ac.theta.grad.clone()
    ac.theta.grad.zero_()
    for sl in (slice(0, cut), slice(cut, B)):
        n = len(data['obs'][sl])
        ac.ppo_minibatch(ac.prepare_batch(data['obs'][sl], data['act'][sl], data['logp'][sl], data['adv'][sl],
                                          data['ret'][sl]), 0.2, 0.5, 0.01, loss_scale=n / B)
    torch.cuda.synchronize()
    assert torch.isfinite(whole).all()
    assert (ac.theta.grad - whole).abs().max().item() < 2e-4 * whole.abs().max().item()


@pytest.mark.parametrize('name', ['cfg3', 'cfg4', 'cfg5'])
def test_masked_oracle_comparison_at_full_size(built_lib, name):
    """The float64 oracle AT BASELINE's full sizes (B = 1024 / 1024 / 2048), through a mask: the samples of a mini-batch are
    independent (ppo.py:36-42: every term of the loss is a mean over samples of per-sample quantities), so the HIP path runs the
    FULL batch -- every size-selected kernel: dw4, sx / pk, the tiled level-0 adjoint, the one-workgroup heads backward, the
    side stream -- with the output adjoint zero outside 16 random samples, and the oracle runs those 16 samples only.  Their
    logp / ent / v must agree to 1e-5 and the full-batch gradient must be the oracle's gradient of the 16 (2e-4 of the slot
    maximum, 1e-2 per entry above 1 % of it): what the other samples leak into the gradient is an error like any other."""
    from tests.helpers import assert_grads, grad_report, make_pair, rel_err
    ac, ref, cfg = make_pair(name, seed=4)
    B = cfg['batch']
    data = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=12)
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(B, size=16, replace=False))
    g = torch.Generator().manual_seed(6)
    w16 = torch.randn(3, 16, generator=g, dtype=torch.float64) * torch.tensor([[1.0], [0.3], [0.7]], dtype=torch.float64)
    w = torch.zeros(3, B, dtype=torch.float64)
    w[:, torch.from_numpy(pick)] = w16
    out = ac.step(data['obs'], data['act'])
    wd = w.cuda()
    (out['logp'].double() * wd[0] + out['ent'].double() * wd[1] + out['v'].double() * wd[2]).sum().backward()
    torch.cuda.synchronize()
    obs16, act16 = [data['obs'][i] for i in pick], np.asarray(data['act'])[pick]
    exp = ref.step(obs16, act16, dtype=torch.float64)
    (exp['logp'] * w16[0] + exp['ent'] * w16[1] + exp['v'] * w16[2]).sum().backward()
    sel = torch.from_numpy(pick)
    for k in ('logp', 'ent', 'v'):
        got = out[k].detach().cpu()[sel]
        assert rel_err(got, exp[k].detach()) < 1e-5, (k, rel_err(got, exp[k].detach()))
    assert torch.isfinite(out['logp']).all() and torch.isfinite(out['v']).all()
    assert_grads(grad_report(ac.theta.grad.detach().double().cpu(), dict(ref.named_parameters()), ac.slot_table))


def _run_worker(name, env_extra, path):
    env = dict(os.environ)
    env.update(env_extra)
    subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'large_worker.py'), name, path], check=True, cwd=ROOT,
                   env=env, timeout=1500)
    return np.load(path)


@pytest.mark.parametrize('name', ['cfg3', 'cfg4', 'cfg5'])
def test_kernel_families_agree_at_full_size(built_lib, tmp_path, name):
    """MFMA GEMM forms + side stream (default) against the VALU forms on one stream, same inputs and weights: the
    predictions, the loss statistics and the whole gradient vector"""
    a = _run_worker(name, {}, str(tmp_path / 'a.npz'))
    # (MG_SX_MIN_ROWS=0: the plain edge-level layout -- the dot block copied into every degree's rows -- which is the one
    # the lane-per-row forms can read; the default run uses the shared block and the shared-input MFMA kernels)
    b = _run_worker(name, {'MG_MFMA': '0', 'MG_MFMA_DX': '0', 'MG_MFMA_DW': '0', 'MG_NO_SIDE_STREAM': '1',
                           'MG_SX_MIN_ROWS': '0'}, str(tmp_path / 'b.npz'))
    assert np.isfinite(a['grad']).all() and np.isfinite(a['pred']).all()
    assert np.abs(a['pred'] - b['pred']).max() < 1e-5 * max(1.0, np.abs(b['pred']).max())
    assert np.abs(a['stats'] - b['stats']).max() < 1e-5 * max(1.0, np.abs(b['stats']).max())
    assert np.abs(a['grad'] - b['grad']).max() < 2e-4 * np.abs(b['grad']).max()


@pytest.mark.parametrize('name', ['cfg4', 'cfg5'])
def test_directional_derivatives_at_full_size(built_lib, name):
    """The hand-written backward against the FORWARD kernels at BASELINE's full sizes, where no oracle reaches: for random
    directions d confined to one parameter slot each, the central difference of L(theta) = sum_b w . (logp, ent, v) along d
    (two forward passes on the HIP path, outputs summed in float64) must equal <dL/dtheta, d> from mg_cov_backward.  A
    float32 forward limits the comparison to a few per cent -- enough to expose a wrong block, a missed accumulation or a
    truncated offset in the size-selected kernels (dw4 / pk / sx, the one-workgroup heads backward), which is its job."""
    cfg = CONFIGS[name]
    ac = _agent(cfg, seed=2)
    B = cfg['batch']
    data = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=9)
    batch = ac.prepare_batch(data['obs'], data['act'])
    g = torch.Generator().manual_seed(4)
    w = torch.randn(3, B, generator=g, dtype=torch.float64).cuda() * torch.tensor([[1.0], [0.3], [0.7]], dtype=torch.float64).cuda()

    def loss():
        with torch.no_grad():
            return (ac.forward_batch(batch).double() * w).sum().item()

    # analytic gradient: the backward entry point with gout = w
    import ctypes as C
    from molgym_amd import _lib
    out = ac.forward_batch(batch)
    ws = ac._last_ws
    grad = torch.zeros_like(ac.theta)
    gout = w.float().contiguous()
    p = lambda t: C.c_void_p(t.data_ptr())
    with ac._guard():
        _lib.check(_lib.lib().mg_cov_backward(C.byref(batch.cfg), p(ac.theta), p(batch.pos), p(batch.charges), p(batch.bags),
                                              p(batch.actions), p(ac.leb), p(ws), ws.numel(), p(gout), p(grad), ac._s()))
    torch.cuda.synchronize()
    assert torch.isfinite(grad).all() and torch.isfinite(out).all()
    slots = [k for k in ac.slot_table if 'weight' in k and any(t in k for t in (
        'input_func_atom', 'rad_funcs.0.linear', 'rad_funcs.2.linear', 'edge_levels.0', 'edge_levels.1', 'edge_levels.2',
        'atom_levels.0', 'atom_levels.1', 'atom_levels.2', 'cg_mix', 'phi_focus', 'phi_trans', 'phi_v', 'phi_d', 'phi_element'))]
    assert len(slots) >= 8, list(ac.slot_table)[:40]
    rng = np.random.default_rng(0)
    picked = [slots[i] for i in rng.choice(len(slots), size=min(10, len(slots)), replace=False)]
    report = {}
    for name_ in picked:
        off, shape = ac.slot_table[name_]
        n = int(np.prod(shape))
        d = torch.zeros_like(ac.theta)
        d[off:off + n] = torch.randn(n, generator=g).cuda()
        rms = ac.theta[off:off + n].detach().pow(2).mean().sqrt().item()
        eps = 2e-2 * max(rms, 1e-3)
        gd = (grad.double() * d.double()).sum().item()
        with torch.no_grad():
            ac.theta.add_(d, alpha=eps)
            lp = loss()
            ac.theta.add_(d, alpha=-2 * eps)
            lm = loss()
            ac.theta.add_(d, alpha=eps)
        fd = (lp - lm) / (2 * eps)
        ref = grad[off:off + n].double().norm().item() * d[off:off + n].double().norm().item()
        report[name_] = (fd, gd, ref)
    bad = {k: v for k, v in report.items() if not abs(v[0] - v[1]) <= 0.05 * abs(v[1]) + 0.01 * v[2]}
    assert not bad, f'finite difference vs analytic directional derivative (fd, <g, d>, |g| |d|): {bad}'
