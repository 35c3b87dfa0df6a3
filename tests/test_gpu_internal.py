"""SchNetAC.step on the HIP path against the CPU oracle (float64): outputs and every parameter gradient."""
import numpy as np
import pytest
import torch

from molgym_amd.spaces import ActionSpace, ObservationSpace
from molgym_amd.synthetic import make_batch_internal
from oracle.internal_ref import SchNetACRef, position_atom
from tests.helpers import abs1_err, rel_err

pytestmark = pytest.mark.gpu
ZS, N = [0, 9, 16], 7


def _pair(seed, width=128, canvas=N):
    from molgym_amd.agents.internal import SchNetAC
    torch.manual_seed(seed)
    ac = SchNetAC(ObservationSpace(canvas, ZS), ActionSpace(ZS), (0.8, 1.8), width, device='cuda:0')
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 1)
        for name, (off, shape) in ac.slot_table.items():
            n = int(np.prod(shape))
            if name.endswith('bias'):
                ac.theta[off:off + n] = (0.1 * torch.randn(n, generator=g)).to(ac.theta)
    ref = SchNetACRef(ZS, canvas, (0.8, 1.8), width).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in ac.export_state_dict().items()}, strict=True)
    return ac, ref


def test_vectorised_zmat_matches_scalar_helper(built_lib):
    from molgym_amd.agents.internal import place_new_atoms
    rng = np.random.default_rng(0)
    B = 40
    natoms = rng.integers(0, N + 1, size=B)
    natoms[:4] = [0, 1, 2, 3]
    pos = rng.normal(size=(B, N, 3)) * 1.5
    focus = np.array([rng.integers(0, max(n, 1)) for n in natoms])
    d, a, h = rng.uniform(0.9, 1.8, B), rng.uniform(0.3, 2.8, B), rng.uniform(-3, 3, B)
    got = place_new_atoms(pos, natoms, focus, d, a, h)
    for b in range(B):
        want = position_atom([pos[b, i] for i in range(natoms[b])], int(focus[b]), d[b], a[b], h[b])
        np.testing.assert_allclose(got[b], want, rtol=1e-12, atol=1e-12)


def test_product_placement_against_the_reference_vectors(built_lib):
    """G8 (tests/golden/g8_zmat.npz: zmat.position_atom_helper of the reference for 0 / 1 / 2 / 3 / 5 atoms, written by
    oracle/make_golden.py): the product's vectorised placement, all five cases in ONE call, exact -- and through the agent's own
    action conversion (SchNetAC.to_action_space), which is what the rollout hands to the environment."""
    import os
    from molgym_amd.agents.internal import place_new_atoms
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g8_zmat.npz'))
    ns = (0, 1, 2, 3, 5)
    pos = np.zeros((len(ns), N, 3))
    args = np.stack([g[f'n{n}_args'] for n in ns])
    for r, n in enumerate(ns):
        pos[r, :n] = g[f'n{n}_pos']
    got = place_new_atoms(pos, np.array(ns), args[:, 0].astype(np.int64), args[:, 1], args[:, 2], args[:, 3])
    np.testing.assert_array_equal(got, np.stack([g[f'n{n}_out'] for n in ns]))


@pytest.mark.parametrize('canvas,width,B', [(7, 128, 24), (12, 128, 24), (20, 128, 24), (7, 64, 24), (7, 128, 140)])
def test_outputs_and_gradients_match_oracle(built_lib, canvas, width, B):
    """(7, 128) and (12, 128): the SchNet interactions as one launch per direction, 8- and 16-atom layouts (schnet_fused.inc);
    (20, 128) and (7, 64): the per-layer launches (molecules above 16 atoms; atom features other than 64); (7, 128, 140):
    BASELINE configs[0] at its real mini-batch size (the sort and the weight-gradient row-chunk classes depend on B)"""
    ac, ref = _pair(0, width, canvas)
    data = make_batch_internal(B, canvas, ZS, seed=4)
    g = torch.Generator().manual_seed(1)
    wl, we, wv = (torch.randn(B, generator=g, dtype=torch.float64) * s for s in (1.0, 0.3, 0.7))
    out = ac.step(data['obs'], data['act'])
    (out['logp'].double() * wl.cuda() + out['ent'].double() * we.cuda() + out['v'].double() * wv.cuda()).sum().backward()
    torch.cuda.synchronize()
    exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    (exp['logp'] * wl + exp['ent'] * we + exp['v'] * wv).sum().backward()
    # (7, 64): the value of a near-empty canvas is ~1e-2 and its float32 error 0.3 - 1.2e-6 absolute from seed to seed with
    # either form of the filter network (tools/dbg_int_err.py): twice the 1e-6 floor of rel_err for that case
    # (7, 128, 140): log-prob is the sum of five head terms of magnitude 1 - 10 each; sample 93 of this batch has them cancel to
    # 0.081, and the float32 error of the sum -- 1.15e-6 absolute, 2e-7 of the terms -- reads as 1.4e-5 of the result
    # (tools/dbg_int_140.py: next worst sample 1.8e-6; the float32 ORACLE is off by 1e-6 on the same sample): 2e-5 as well
    tol = 2e-5 if (width == 64 or B == 140) else 1e-5
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < tol, (k, rel_err(out[k], exp[k]))
    got = ac.theta.grad.double().cpu()
    want = dict(ref.named_parameters())
    bad = {}
    for name, (off, shape) in ac.slot_table.items():
        n = int(np.prod(shape))
        gw = want[name].grad
        gw = torch.zeros(n, dtype=torch.float64) if gw is None else gw.reshape(-1)
        scale = gw.abs().max().item()
        err = (got[off:off + n] - gw).abs().max().item() / max(scale, 1e-12)
        if not (err < 2e-4 or scale < 1e-10):
            bad[name] = (err, scale)
    assert not bad, bad


def test_evaluate_actions_matches_oracle(built_lib):
    """the north_star's spelling of action evaluation (agents/base.py:17-19) on the internal-coordinate agent"""
    ac, ref = _pair(5)
    data = make_batch_internal(13, N, ZS, seed=2)
    with torch.no_grad():
        out = ac.evaluate_actions(data['obs'], data['act'])
        exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k


def test_graph_step_equals_stream_launches(built_lib):
    """mg_int_ppo_step as one updated hipGraph launch == the same ~58 launches issued to the stream (statistics and step outputs
    bit for bit, gradient to the reproducibility of its float atomics), for mini-batches of different ragged sizes through the
    same cached graph"""
    ac, ref = _pair(6)
    for k, B in enumerate((20, 33, 7, 140)):  # (140: BASELINE configs[0]'s mini-batch, what bench.py --agent internal times)
        d = make_batch_internal(B, N, ZS, seed=10 + k)
        batch = ac.prepare_batch(d['obs'], d['act'], d['logp'], d['adv'], d['ret'])
        res = {}
        for mode in (False, True):
            ac.theta.grad = torch.zeros_like(ac.theta)
            acc = torch.zeros(6, dtype=torch.float64, device='cuda')
            stats = ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, loss_scale=0.25, stats_accum=acc, graph=mode)
            torch.cuda.synchronize()
            assert ac.last_step_used_graph == mode
            res[mode] = (stats.clone(), ac.theta.grad.clone(), acc.clone())
        assert torch.equal(res[False][0], res[True][0])
        assert torch.allclose(res[True][2], 0.25 * res[True][0], rtol=1e-14, atol=0)
        assert (res[False][1] - res[True][1]).abs().max().item() <= 2e-5 * res[False][1].abs().max().item()


def test_epoch_cache_equals_per_minibatch_preparation(built_lib):
    """[r5] the derived weight matrices once per EPOCH (mg_int_ppo_step flags = MG_STEP_WEIGHTS_CURRENT; they sit first in the
    workspace, at offsets that do not depend on the batch) == every mini-batch preparing them itself: statistics bit for bit, the
    epoch's gradient to the reproducibility of its float atomics, for mini-batches of different ragged sizes through ONE cached
    workspace, graph and stream form, and across a change of theta (ppo.py:117-146: theta is constant over an epoch)"""
    ac, ref = _pair(8)
    batches = []
    for k, B in enumerate((24, 31, 9, 140, 24)):
        d = make_batch_internal(B, N, ZS, seed=40 + k)
        batches.append(ac.prepare_batch(d['obs'], d['act'], d['logp'], d['adv'], d['ret']))
    for graph in (True, False):
        for epoch in range(2):
            res = {}
            for cached in (False, True):
                ac.theta.grad = torch.zeros_like(ac.theta)
                ac.invalidate_weights()
                outs = [ac.ppo_minibatch(b, 0.2, 0.5, 0.01, loss_scale=0.5, graph=graph, epoch_cache=cached).clone() for b in batches]
                if cached:
                    assert ac._ws_epoch[0]['weights']
                    ac.fold_gradients()
                torch.cuda.synchronize()
                res[cached] = (outs, ac.theta.grad.clone())
            for s0, s1 in zip(res[False][0], res[True][0]):
                assert torch.equal(s0, s1)
            g0, g1 = res[False][1], res[True][1]
            assert torch.isfinite(g1).all() and (g0 - g1).abs().max().item() <= 2e-5 * g0.abs().max().item()
            with torch.no_grad():  # "the optimizer steps": the next epoch must see the new theta
                ac.theta.add_(0.01 * torch.randn_like(ac.theta))


def test_small_canvases_and_masks(built_lib):
    """n = 0, 1, 2 exercise the action masks (distance / angle / dihedral / kappa) and the null-atom focus."""
    ac, ref = _pair(3, width=64)
    data = make_batch_internal(12, N, ZS, seed=7)
    obs = list(data['obs'])
    for b, keep in enumerate((0, 1, 2, 3)):
        canvas, bag = obs[1]  # a full canvas
        canvas = tuple(item if i < keep else (0, (0.0, 0.0, 0.0)) for i, item in enumerate(canvas))
        obs[b] = (canvas, bag)
        data['act'][b, 1] = 0
    with torch.no_grad():
        out = ac.step(obs, data['act'])
        exp = ref.step(obs, data['act'], dtype=torch.float64)
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k


def test_rollout_sampling_is_consistent_with_evaluation(built_lib):
    """step(obs) draws a full 7-column action row per sample; evaluating the drawn rows with step(obs, actions)
    and with the float64 oracle reproduces the logp / ent / v the sampling call reported."""
    ac, ref = _pair(5, width=64)
    data = make_batch_internal(24, N, ZS, seed=11)
    obs = list(data['obs'])
    canvas, bag = obs[1]
    obs[0] = (tuple((0, (0.0, 0.0, 0.0)) for _ in canvas), bag)  # an empty canvas: focus must be slot 0
    for training in (True, False):
        ac.training = training
        torch.manual_seed(123)
        out = ac.step(obs)
        a = out['a'].cpu().numpy()
        assert a.shape == (len(obs), 7) and np.all(a[:, 0] == 0)
        natoms = np.array([sum(1 for it in o[0] if ZS[it[0]] != 0) for o in obs])
        assert np.all(a[:, 1] < np.maximum(natoms, 1)) and a[0, 1] == 0
        for b, (_, bg) in enumerate(obs):
            assert bg[int(a[b, 2])] > 0  # only elements left in the bag are drawn
        assert np.all(a[:, 3] >= 0.001) and set(np.unique(a[:, 6])) <= {0.0, 1.0}
        with torch.no_grad():
            again = ac.step(obs, a)
            exp = ref.step(obs, a, dtype=torch.float64)
        for k in ('logp', 'ent', 'v'):
            assert torch.equal(out[k], again[k]), k
            assert abs1_err(out[k], exp[k]) < 1e-5, k  # drawn actions: values near zero occur (absolute below one)
        assert len(out['actions']) == len(obs)
        idx, pos = out['actions'][1]
        assert 0 <= idx < len(ZS) and len(pos) == 3
        # the returned position is the z-matrix placement of the drawn row (kappa = 1 flips the dihedral)
        atoms = [np.asarray(x, dtype=np.float64) for l, x in obs[1][0] if ZS[l] != 0]
        sign = -1.0 if a[1, 6] else 1.0
        want = position_atom(atoms, int(a[1, 1]), float(a[1, 3]), float(a[1, 4]), sign * float(a[1, 5]))
        assert np.allclose(pos, want, atol=1e-9)
    # evaluation mode is deterministic; training mode follows torch's RNG
    ac.training = True
    torch.manual_seed(7)
    a1 = ac.step(obs)['a'].cpu().numpy()
    torch.manual_seed(7)
    a2 = ac.step(obs)['a'].cpu().numpy()
    torch.manual_seed(8)
    a3 = ac.step(obs)['a'].cpu().numpy()
    assert np.array_equal(a1, a2) and not np.array_equal(a1, a3)


def test_sampled_focus_frequencies_follow_the_softmax(built_lib):
    """many draws for one canvas: empirical focus / kappa frequencies match the distribution parameters read back"""
    ac, _ = _pair(9, width=64)
    data = make_batch_internal(8, N, ZS, seed=3)
    full = max(data['obs'], key=lambda o: sum(1 for it in o[0] if ZS[it[0]] != 0))
    obs = [full] * 512
    ac.training = True
    torch.manual_seed(0)
    out = ac.step(obs)
    a = out['a'].cpu().numpy()
    n = sum(1 for it in full[0] if ZS[it[0]] != 0)
    batch = ac.make_batch(obs[:1], a[:1])
    _, ws = ac._forward_nograd(batch)
    p = torch.softmax(ac._ws_view(batch.cfg, ws, 'logitF')[:n], 0).cpu().numpy()
    freq = np.bincount(a[:, 1].astype(int), minlength=n)[:n] / len(obs)
    assert np.abs(freq - p).max() < 4.0 * np.sqrt(0.25 / len(obs))


def test_ppo_train_with_sampled_rollout_tracks_the_oracle_loop(built_lib):
    """rollout rows drawn by SchNetAC.step(obs), then two epochs of molgym_amd.ppo.train (generic autograd path)
    against the same loop on the float64 oracle with torch Adam"""
    from molgym_amd import ppo
    from oracle.ppo_ref import batch_indices_ref, compute_loss_ref
    ac, ref = _pair(12, width=64)
    data = make_batch_internal(24, N, ZS, seed=21)
    ac.training = True
    torch.manual_seed(1)
    drawn = ac.step(list(data['obs']))
    data['act'] = drawn['a'].cpu().numpy().astype(np.float64)
    data['logp'] = drawn['logp'].detach().double().cpu().numpy() - 0.002
    opt = torch.optim.Adam(ac.parameters(), lr=3e-4)
    ropt = torch.optim.Adam(ref.parameters(), lr=3e-4)
    np.random.seed(5)
    infos = ppo.train(ac, opt, data, mini_batch_size=10, clip_ratio=0.2, target_kl=1e9, vf_coef=0.5, entropy_coef=0.01,
                      gradient_clip=0.5, max_num_steps=2)
    np.random.seed(5)
    for _ in range(2):
        ropt.zero_grad()
        stats = []
        for idx in batch_indices_ref(24, 10):
            sub = ppo.collect_data_batch(data, idx)
            loss, info = compute_loss_ref(ref, sub, 0.2, 0.5, 0.01, step_dtype=torch.float64)
            loss.backward()
            stats.append(info)
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        ropt.step()
    assert infos['num_opt_steps'] == 2
    mean = ppo.compute_mean_dict(stats)
    for k in ppo.KEYS:
        assert abs(infos[k] - mean[k]) < 2e-4 * max(1.0, abs(mean[k])), (k, infos[k], mean[k])
    new = ac.export_state_dict()
    for k, v in ref.state_dict().items():
        d = (new[k].double().cpu() - v).abs().max().item()
        assert d < 1e-4, (k, d)


def test_device_minibatch_equals_autograd_path(built_lib):
    """SchNetAC.ppo_minibatch (forward + float64 loss kernel + backward, no autograd graph) against
    molgym_amd.ppo.compute_loss + loss.backward() through the autograd Function"""
    from molgym_amd import ppo
    ac, _ = _pair(14, width=64)
    data = make_batch_internal(20, N, ZS, seed=31)
    loss, info = ppo.compute_loss(ac, data, clip_ratio=0.2, vf_coef=0.5, entropy_coef=0.01)
    ac.theta.grad = None
    loss.backward()
    want = ac.theta.grad.clone()
    ac.theta.grad = None
    stats = ac.ppo_minibatch(ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret']),
                             0.2, 0.5, 0.01)
    torch.cuda.synchronize()
    assert (ac.theta.grad - want).abs().max().item() < 1e-5 * want.abs().max().item()
    assert abs(stats[0].item() - info['policy_loss']) < 1e-6 * max(1.0, abs(info['policy_loss']))
