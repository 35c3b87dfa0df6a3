"""Forward parity of the HIP CovariantAC.step against the CPU oracle (float64), stage by stage."""
import numpy as np
import pytest
import torch

from molgym_amd.synthetic import make_batch
from tests.helpers import abs1_err, compact_edges, compact_vec, make_pair, rel_err

pytestmark = pytest.mark.gpu


def _run(cfg_name='cfg2', B=24, seed=0, beta='cfg'):
    ac, ref, cfg = make_pair(cfg_name, seed=seed, beta=beta)
    data = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=seed + 3)
    with torch.no_grad():
        out = ac.step(data['obs'], data['act'])
        exp = ref.step(data['obs'], data['act'], dtype=torch.float64, return_internals=True)
    torch.cuda.synchronize()
    return ac, ref, cfg, data, out, exp


def test_outputs_match_oracle(built_lib):
    ac, ref, cfg, data, out, exp = _run()
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k


def test_evaluate_actions_matches_oracle(built_lib):
    """`agent.evaluate_actions(obs, actions)` -- the name BASELINE.json's north_star uses for action evaluation
    (/root/reference/molgym/agents/base.py:17-19 spells it step(observations, actions); ppo.py:26 is its caller) -- against the
    oracle: outputs to 1e-5 and, through its autograd node, every parameter gradient."""
    from tests.helpers import assert_grads, grad_report
    ac, ref, cfg = make_pair('cfg2', seed=21)
    data = make_batch(17, cfg['canvas_size'], cfg['zs'], seed=9)
    out = ac.evaluate_actions(data['obs'], data['act'])
    exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k
    assert torch.equal(out['a'].cpu(), torch.as_tensor(data['act'], dtype=torch.float32))
    assert len(out['dists']) == 4  # focus, element, distance, orientation (covariant/agent.py:318-334)
    (0.3 * out['logp'].sum() + out['v'].sum() - 0.02 * out['ent'].sum()).backward()
    (0.3 * exp['logp'].sum() + exp['v'].sum() - 0.02 * exp['ent'].sum()).backward()
    assert_grads(grad_report(ac.theta.grad, dict(ref.named_parameters()), ac.slot_table))
    # and it IS the training-path entry: same numbers as step(obs, actions)
    with torch.no_grad():
        again = ac.step(data['obs'], data['act'])
    for k in ('logp', 'ent', 'v'):
        assert torch.equal(out[k].detach(), again[k]), k


def test_outputs_match_oracle_so3_without_beta(built_lib):
    ac, ref, cfg, data, out, exp = _run(beta=None, seed=2)
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k


def test_outputs_match_oracle_five_elements(built_lib):
    ac, ref, cfg, data, out, exp = _run('cfg3', B=16, seed=1)
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k


def test_encoder_stages(built_lib):
    """Every saved intermediate of the encoder against the oracle's, to localise a deviation."""
    ac, ref, cfg, data, out, exp = _run(B=12, seed=5)
    from oracle.covariant_ref import parse_observations
    d = parse_observations(data['obs'], cfg['zs'], cfg['canvas_size'], torch.float64)
    with torch.no_grad():
        atoms_all, edges_all, extra = ref.cg_model(d, return_all=True)
    am, em = d['atom_mask'], d['edge_mask']
    natoms = d['num_atoms'].numpy()
    ccfg = ac._make_cfg(len(data['obs']), natoms)
    report = {}

    def chk(name, want, floor=1e-2):
        got = ac.workspace_view(name, ccfg)[:want.numel()].view(want.shape).double().cpu()
        report[name] = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-30)

    chk('r', extra['norms'][em])
    TE = int(em.sum())
    y_want = torch.cat([p[em].reshape(TE, -1) for p in extra['sph']], dim=1)
    chk('Y', y_want)
    chk('A0', extra['atom_in'][0][am].reshape(int(am.sum()), -1))
    for k in range(3):
        a_parts = compact_vec(atoms_all[k], am)
        for l in range(5):
            chk(f'A{k + 1}_{l}', a_parts[l])
    e_last = compact_edges(edges_all[2], em)
    for l in range(5):
        chk(f'Elast_{l}', e_last[l])
    chk('inv', exp['invariats'][am])
    bad = {k: v for k, v in report.items() if not v < 1e-5}
    assert not bad, f'stages off: {bad}; all: {report}'


def test_head_parts(built_lib):
    ac, ref, cfg, data, out, exp = _run(B=20, seed=7)
    natoms = exp['data']['num_atoms'].numpy()
    ccfg = ac._make_cfg(len(data['obs']), natoms)
    B = len(data['obs'])
    parts = ac.workspace_view('parts', ccfg).view(6, B)
    names = ['focus', 'element', 'distance', 'so3']
    for i, n in enumerate(names):
        assert abs1_err(parts[i], exp['logps'][i]) < 1e-5, n  # one head's share of logp: sums of O(1) terms, may be ~0
    assert rel_err(parts[4], exp['ent_parts'][0], floor=1e-3, abs_tol=1e-7, tol=1e-4) < 1e-4
    assert rel_err(parts[5], exp['ent_parts'][1], floor=1e-3, abs_tol=1e-7, tol=1e-4) < 1e-4
    assert rel_err(ac.workspace_view('logz', ccfg), exp['log_z']) < 1e-5


def test_empty_and_full_canvases(built_lib):
    """n = 0 (empty canvas: focus on slot 0, zero covariants) and n = N in the same batch."""
    ac, ref, cfg = make_pair('cfg2', seed=11)
    data = make_batch(6, cfg['canvas_size'], cfg['zs'], seed=4)  # make_batch forces counts[0]=0, counts[1]=N
    with torch.no_grad():
        out = ac.step(data['obs'], data['act'])
        exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k


def test_all_empty_batch(built_lib):
    ac, ref, cfg = make_pair('cfg2', seed=12)
    data = make_batch(3, cfg['canvas_size'], cfg['zs'], seed=5)
    N = cfg['canvas_size']
    empty = tuple([(0, (0.0, 0.0, 0.0))] * N)
    obs = [(empty, o[1]) for o in data['obs']]
    act = data['act'].copy()
    act[:, 0] = 0
    with torch.no_grad():
        out = ac.step(obs, act)
        exp = ref.step(obs, act, dtype=torch.float64)
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k


def test_bad_indices_raise(built_lib):
    ac, ref, cfg = make_pair('cfg2', seed=13)
    data = make_batch(4, cfg['canvas_size'], cfg['zs'], seed=6)
    act = data['act'].copy()
    act[0, 0] = cfg['canvas_size']  # focus out of range -> to_one_hot raises in the reference
    with pytest.raises(RuntimeError):
        ac.step(data['obs'], act)
    act = data['act'].copy()
    act[1, 1] = len(cfg['zs'])
    with pytest.raises(RuntimeError):
        ac.step(data['obs'], act)


def test_whole_module_pickle_round_trip(built_lib, tmp_path):
    """ModelIO saves the whole nn.Module with torch.save (tools/model_util.py:82-100); the agent must survive it."""
    ac, ref, cfg = make_pair('cfg2', seed=14)
    data = make_batch(5, cfg['canvas_size'], cfg['zs'], seed=2)
    with torch.no_grad():
        before = ac.step(data['obs'], data['act'])['logp'].cpu()
    path = str(tmp_path / 'agent.model')
    torch.save(ac, path)
    assert __import__('os').path.getsize(path) < 4 * 1024 * 1024  # parameters only, no workspace
    again = torch.load(path, weights_only=False)
    again.observation_space, again.action_space = ac.observation_space, ac.action_space  # run.py:53-54
    with torch.no_grad():
        after = again.step(data['obs'], data['act'])['logp'].cpu()
    assert torch.equal(before, after)


@pytest.mark.parametrize('zs,N,B,kw', [
    ([0, 1, 6, 7, 8, 9, 16, 17], 6, 9, {}),        # Z = 8, the ABI maximum (Co = 32, 384 invariants per atom)
    ([0, 9], 1, 5, {}),                             # canvas of a single slot
    ([0, 1, 6], 16, 3, {}),                         # N = 16
    ([0, 9, 16], 7, 1, {}),                         # a mini-batch of one
    ([0, 9, 16], 7, 11, {'network_width': 64}),     # narrower heads
])
def test_shape_corners_match_oracle(built_lib, zs, N, B, kw):
    """outputs and every parameter gradient at the corners of the supported shape space"""
    from molgym_amd.agents.covariant import CovariantAC
    from molgym_amd.spaces import ActionSpace, ObservationSpace
    from molgym_amd.synthetic import MODEL_DEFAULTS, make_batch
    from oracle.covariant_ref import CovariantACRef
    md = dict(MODEL_DEFAULTS)
    md.update(kw)
    torch.manual_seed(len(zs) * 100 + N)
    ac = CovariantAC(ObservationSpace(N, zs), ActionSpace(zs), bag_scale=5, beta=-10.0, device='cuda:0', **md)
    ref = CovariantACRef(zs=zs, canvas_size=N, bag_scale=5, beta=-10.0, **md).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in ac.export_state_dict().items()})
    d = make_batch(B, N, zs, seed=N)
    out = ac.step(d['obs'], d['act'])
    (out['logp'].sum() * 0.3 + out['v'].sum() - 0.02 * out['ent'].sum()).backward()
    exp = ref.step(d['obs'], d['act'], dtype=torch.float64)
    (exp['logp'].sum() * 0.3 + exp['v'].sum() - 0.02 * exp['ent'].sum()).backward()
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k], exp[k]) < 1e-5, k
    want = dict(ref.named_parameters())
    for k, (o, shp) in ac.slot_table.items():
        n = int(np.prod(shp))
        w = want[k].grad.reshape(-1)
        g = ac.theta.grad[o:o + n].double().cpu()
        assert (g - w).abs().max().item() <= 2e-4 * max(w.abs().max().item(), 1e-3), k


def test_unsupported_width_is_rejected(built_lib):
    from molgym_amd.agents.covariant import CovariantAC
    from molgym_amd.spaces import ActionSpace, ObservationSpace
    from molgym_amd.synthetic import MODEL_DEFAULTS
    md = dict(MODEL_DEFAULTS)
    # (256 is supported since round 5 -- the staged heads, tests/test_gpu_parity_full.py::test_network_width_256_vs_oracle; what is
    # rejected: widths that are not a multiple of 4, and widths above 1024)
    md['network_width'] = 130
    with pytest.raises(RuntimeError):
        CovariantAC(ObservationSpace(7, [0, 9, 16]), ActionSpace([0, 9, 16]), bag_scale=5, beta=-10.0, device='cuda:0', **md)
    md['network_width'] = 2048
    with pytest.raises(RuntimeError):
        CovariantAC(ObservationSpace(7, [0, 9, 16]), ActionSpace([0, 9, 16]), bag_scale=5, beta=-10.0, device='cuda:0', **md)
