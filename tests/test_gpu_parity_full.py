"""Oracle parity at the sizes BASELINE.json names: canvases of 20 and 40 slots with five elements (the neighbour-tile
loops of the CG kernels, the two-kernel list build and the 64-row GEMM forms only run at n > 16) and the SF6
mini-batch at its real size of 140 samples (counting sort of the atoms, dW row-chunk classes depend on B / TA).
Outputs to 1e-5 relative (helpers.rel_err: true relative above 1e-2, absolute 1e-6 below), EVERY parameter gradient to 2e-4 of
its per-tensor maximum AND to 1e-2 relative on every entry above 1 % of that maximum, against the float64 oracle."""
import numpy as np
import pytest
import torch

from molgym_amd.synthetic import make_batch
from tests.helpers import assert_grads, grad_report, make_pair, oracle_backward, rel_err

pytestmark = pytest.mark.gpu


def _compare(cfg_name, B, seed, weights=(1.0, 0.3, 0.7), data=None):
    ac, ref, cfg = make_pair(cfg_name, seed=seed)
    data = data or make_batch(B, cfg['canvas_size'], cfg['zs'], seed=seed + 3)
    B = len(data['obs'])
    g = torch.Generator().manual_seed(seed)
    wl, we, wv = (torch.randn(B, generator=g, dtype=torch.float64) * s for s in weights)
    out = ac.step(data['obs'], data['act'])
    (out['logp'].double() * wl.cuda() + out['ent'].double() * we.cuda() + out['v'].double() * wv.cuda()).sum().backward()
    torch.cuda.synchronize()
    exp, want = oracle_backward(ref, data, (wl, we, wv))
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k].detach(), exp[k].detach()) < 1e-5, (k, rel_err(out[k].detach(), exp[k].detach()))
    got = ac.theta.grad.detach().double().cpu()
    assert_grads(grad_report(got, want, ac.slot_table))
    return ac, cfg, data


def _crowded(cfg_name, counts, seed):
    """make_batch draws U{0..N} atoms; these canvases are (nearly) full so that every neighbour tile is populated"""
    from molgym_amd.synthetic import CONFIGS, make_canvas
    cfg = CONFIGS[cfg_name]
    rng = np.random.default_rng(seed)
    d = make_batch(len(counts), cfg['canvas_size'], cfg['zs'], seed=seed)
    obs = []
    for b, n in enumerate(counts):
        obs.append((make_canvas(rng, n, cfg['canvas_size'], len(cfg['zs'])), d['obs'][b][1]))
        d['act'][b, 0] = rng.integers(0, max(n, 1))
    d['obs'] = obs
    return d


@pytest.mark.parametrize('seed', [0, 1])
def test_canvas20_five_elements_vs_oracle(built_lib, seed):
    _compare('cfg4', 4, seed)


def test_canvas20_crowded_vs_oracle(built_lib):
    _compare('cfg4', 3, 7, data=_crowded('cfg4', [20, 17, 19], 7))


@pytest.mark.parametrize('seed', [0, 1])
def test_canvas40_five_elements_vs_oracle(built_lib, seed):
    _compare('cfg5', 3, seed)


def test_canvas40_crowded_vs_oracle(built_lib):
    _compare('cfg5', 2, 5, data=_crowded('cfg5', [40, 33], 5))


def test_canvas12_multibag_vs_oracle(built_lib):
    _compare('cfg3', 24, 2)


def test_sf6_full_minibatch_140_vs_oracle(built_lib):
    """cfg2 at the batch size of the headline metric"""
    ac, cfg, data = _compare('cfg2', 140, 0)
    # and the list build reported no inconsistency (encoder.inc: err = 1 not compacted, 2 = TA / TE mismatch)
    from oracle.covariant_ref import parse_observations
    natoms = parse_observations(data['obs'], cfg['zs'], cfg['canvas_size'], torch.float64)['num_atoms'].numpy()
    assert int(ac.workspace_view_int('err', ac._make_cfg(140, natoms))[0]) == 0


# ---- kernels that only run at sizes the oracle does not reach in seconds: forced on the small oracle cases --------------------------
# Their switches are read once per process, hence child interpreters.  ALL children are started together by whichever of the
# tests below runs first (round 5 ran them one after the other: 254 s of a 635 s suite); they share the GPU with the parent's
# own tests and the oracle results of a case are computed once per session (tests/helpers.py::oracle_backward: disk cache
# keyed by the weights, the inputs and the oracle's sources).
_FORCED = {
    # With many edges (>= 16384) the edge levels keep ONE copy of the DotMatrix block and run the shared-input MFMA kernels
    # (gemm.inc: k_gemm_mfma_sx / k_gemm_mfma_pk, segmented weight-gradient inputs): MG_SX_MIN_ROWS=1; MG_SX_WS=0: the
    # wave-per-16-rows kernel, MG_SX_WS=2: the LDS-stationary-weights kernel of the large row counts (round 5)
    'sx0': (dict(MG_SX_MIN_ROWS='1', MG_SX_WS='0'), ('parity_full', 'backward', 'forward'),
            'canvas20_crowded or canvas40_crowded or sf6_full or grads_cfg2 or grads_five or encoder_stages'),
    'sx2': (dict(MG_SX_MIN_ROWS='1', MG_SX_WS='2'), ('parity_full', 'backward', 'forward'),
            'canvas20_crowded or canvas40_crowded or sf6_full or grads_cfg2 or grads_five or encoder_stages'),
    # The adjoints of the atom cat-mixes w.r.t. their concatenated inputs run a weight-stationary kernel from 4096 row tiles
    # (gemm.inc: k_gemm_mfma_cols_ws): R = 20 / 24 / 40 (hidden levels; last level of Z = 3 and Z = 5), partial last row tiles,
    # one row-tile chunk per workgroup
    'cols_ws': (dict(MG_COLS_WS_MIN_TILES='1', MG_COLS_WS_WGS='64'), ('parity_full', 'backward'),
                'canvas20_crowded or canvas40_crowded or sf6_full or grads_cfg2 or grads_five or canvas12'),
    # [r5] the LDS-stationary-weights row GEMM of the atom cat-mixes (gemm.inc: k_gemm_mfma_rows_ws; MG_ROWS_WS=2 also takes the
    # 40-wide last-level mixes of Z = 5) and the opt-in molecule-stationary CG adjoint (backward.inc: k_catbuild_bwd_mol)
    'r5_large': (dict(MG_ROWS_WS_MIN='1', MG_ROWS_WS='2', MG_CGB_MOL='0'), ('parity_full', 'backward'),
                 'canvas20_crowded or canvas40_crowded or sf6_full or grads_five'),
    # [r5] MG_CATMIX_EPI=1: the atom cat-mix of the levels >= 1 as the epilogue of the CG kernel (cg_mfma.inc: CgMix) instead of
    # the row GEMM launch -- off by default (measured neutral)
    'catmix_epi': (dict(MG_CATMIX_EPI='1'), ('parity_full', 'backward', 'forward'),
                   'sf6_full or grads_cfg2 or encoder_stages or canvas12'),
    # [r6] the two older placements of the PPO loss (MG_FUSED_LOSS=2: coefficients and statistics in the last workgroup of
    # k_heads_fwd; 0: the loss as its own launch) against the same oracle / autograd comparisons the default placement runs
    'loss_in_fwd_tail': (dict(MG_FUSED_LOSS='2'), ('ppo', 'internal'), 'fused_minibatch or ppo_loss_kernel or graph_step or ppo_minibatch'),
    'loss_own_launch': (dict(MG_FUSED_LOSS='0'), ('ppo', 'internal'), 'fused_minibatch or ppo_loss_kernel or graph_step or ppo_minibatch'),
    # [r6] SchNetAC's head chains as grouped GEMM / gather / scatter launches (internal.inc; the default up to 16 atoms is one
    # launch per direction, int_heads_fused.inc -- canvas 20 of the default run already takes the staged form)
    'int_heads_staged': (dict(MG_INT_HEADS_FUSED='0'), ('internal',),
                         'outputs_and_gradients or graph_step or epoch_cache or small_canvases or device_minibatch'),
    # ... and the one-launch form WITHOUT the weights requested at the top (k_int_heads_fwd / _bwd: the kernels of the shapes
    # k_int_heads_*_pre do not take, forced on the default shapes)
    'int_heads_plain': (dict(MG_INT_HEADS_FUSED='2'), ('internal',),
                        'outputs_and_gradients or graph_step or epoch_cache or small_canvases or device_minibatch'),
}
_CHILD = {}


def _forced_child(name):
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    if not _CHILD:
        for nm, (env, files, kexpr) in _FORCED.items():
            log = tempfile.NamedTemporaryFile('w+', prefix=f'forced_{nm}_', suffix='.log', delete=False)
            cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider'] + \
                  [os.path.join(here, f'test_gpu_{f}.py') for f in files] + ['-k', kexpr]
            _CHILD[nm] = (subprocess.Popen(cmd, env=dict(os.environ, **env), stdout=log, stderr=subprocess.STDOUT, text=True), log)
    proc, log = _CHILD[name]
    try:
        rc = proc.wait(timeout=1500)
    except subprocess.TimeoutExpired:
        proc.kill()
        raise
    log.seek(0)
    text = log.read()
    assert rc == 0, f'{name} {_FORCED[name][0]}: ' + text[-4000:]
    assert ' passed' in text and 'deselected' in text, text[-2000:]


@pytest.mark.parametrize('sx_ws', ['sx0', 'sx2'])
def test_shared_dot_block_layout_vs_oracle(built_lib, sx_ws):
    _forced_child(sx_ws)


def test_weight_stationary_column_gemm_vs_oracle(built_lib):
    _forced_child('cols_ws')


def test_round5_forced_large_batch_kernels_vs_oracle(built_lib):
    _forced_child('r5_large')


def test_catmix_epilogue_vs_oracle(built_lib):
    _forced_child('catmix_epi')


@pytest.mark.parametrize('variant', ['loss_in_fwd_tail', 'loss_own_launch'])
def test_other_loss_placements(built_lib, variant):
    _forced_child(variant)


@pytest.mark.parametrize('variant', ['int_heads_staged', 'int_heads_plain'])
def test_internal_agent_other_head_kernels_vs_oracle(built_lib, variant):
    _forced_child(variant)


def test_other_channel_counts_vs_oracle(built_lib):
    """num_channels_hidden = 8, num_channels_per_element = 2 (arg_parser.py:55-60 makes them command-line flags): the channel
    counts are compile-time constants of a library build, so this agent loads its own build of the same sources
    (molgym_amd/_lib.py::build_variant; __graft_entry__.build() pre-builds this one) -- outputs and every parameter
    gradient against the oracle, plus the C-side parameter layout against molgym_amd/layout.py"""
    import ctypes as C
    from molgym_amd import _lib, layout
    ac, ref, cfg = make_pair('cfg2', seed=21, num_channels_hidden=8, num_channels_per_element=2)
    lib = ac._L()
    ch, ce = C.c_int32(), C.c_int32()
    lib.mg_cov_channels(C.byref(ch), C.byref(ce))
    assert (ch.value, ce.value) == (8, 2) and lib is not _lib.lib()
    ccfg = ac._make_cfg(1, np.array([1]))
    n = C.c_int64()
    _lib.check(lib.mg_cov_num_params(C.byref(ccfg), C.byref(n)), lib)
    table, total = layout.offsets(len(cfg['zs']), 128, 3, 8, 2)
    assert n.value == total == ac.theta.numel() and sum(p.numel() for p in ref.parameters()) == total
    data = make_batch(12, cfg['canvas_size'], cfg['zs'], seed=33)
    B = len(data['obs'])
    g = torch.Generator().manual_seed(2)
    wl, we, wv = (torch.randn(B, generator=g, dtype=torch.float64) * s for s in (1.0, 0.3, 0.7))
    out = ac.step(data['obs'], data['act'])
    (out['logp'].double() * wl.cuda() + out['ent'].double() * we.cuda() + out['v'].double() * wv.cuda()).sum().backward()
    torch.cuda.synchronize()
    exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    (exp['logp'] * wl + exp['ent'] * we + exp['v'] * wv).sum().backward()
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k].detach(), exp[k].detach()) < 1e-5, (k, rel_err(out[k].detach(), exp[k].detach()))
    assert_grads(grad_report(ac.theta.grad.detach().double().cpu(), dict(ref.named_parameters()), ac.slot_table))


def test_network_width_256_vs_oracle(built_lib):
    """[r5] --network_width above 128 (a flag upstream, arg_parser.py:57; every BASELINE config uses 128): the heads run as the
    staged kernels + row GEMMs of any width (heads_fused.inc::use_staged_heads) instead of the one-launch-per-direction form --
    outputs and every parameter gradient against the oracle built with the same width, and the one-call PPO step against autograd"""
    ac, ref, cfg = make_pair('cfg2', seed=23, network_width=256)
    data = make_batch(24, cfg['canvas_size'], cfg['zs'], seed=37)
    B = len(data['obs'])
    g = torch.Generator().manual_seed(4)
    wl, we, wv = (torch.randn(B, generator=g, dtype=torch.float64) * s for s in (1.0, 0.3, 0.7))
    out = ac.step(data['obs'], data['act'])
    (out['logp'].double() * wl.cuda() + out['ent'].double() * we.cuda() + out['v'].double() * wv.cuda()).sum().backward()
    torch.cuda.synchronize()
    exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    (exp['logp'] * wl + exp['ent'] * we + exp['v'] * wv).sum().backward()
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k].detach(), exp[k].detach()) < 1e-5, (k, rel_err(out[k].detach(), exp[k].detach()))
    assert_grads(grad_report(ac.theta.grad.detach().double().cpu(), dict(ref.named_parameters()), ac.slot_table))
    # the PPO mini-batch call (forward, float64 loss, backward in one call) == loss.backward() through the autograd node
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    ac.theta.grad = torch.zeros_like(ac.theta)
    ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
    torch.cuda.synchronize()
    g_dev = ac.theta.grad.clone()
    ac.theta.grad = None
    from molgym_amd import ppo as hip_ppo
    loss, _ = hip_ppo.compute_loss(ac, data, 0.2, 0.5, 0.01)
    loss.backward()
    torch.cuda.synchronize()
    assert (g_dev - ac.theta.grad).abs().max().item() <= 2e-4 * ac.theta.grad.abs().max().item()


@pytest.mark.parametrize('maxl', [3, 2])
def test_other_maxl_vs_oracle(built_lib, maxl):
    """[r6] --maxl 3 / 2 (arg_parser.py:56; 4 is the default and what every BASELINE config uses): theta and state_dict have the
    maxl-limited module's shapes, the kernels read its embedding into the maxl = 4 layout (CovariantAC._ktheta,
    layout.embedding_index; exactness of the embedding itself: tests/test_host.py).  Outputs and every parameter gradient
    against the oracle BUILT WITH THE SAME maxl, the PPO mini-batch call (one graph launch, gradient gathered back) against the
    autograd path, the epoch-cached form of it, and the rollout's sampling launch."""
    ac, ref, cfg = make_pair('cfg2', seed=29, maxl=maxl)
    assert ac.theta.numel() == sum(p.numel() for p in ref.parameters()) < 185006
    data = make_batch(12, cfg['canvas_size'], cfg['zs'], seed=39)
    B = len(data['obs'])
    g = torch.Generator().manual_seed(5)
    wl, we, wv = (torch.randn(B, generator=g, dtype=torch.float64) * s for s in (1.0, 0.3, 0.7))
    out = ac.step(data['obs'], data['act'])
    (out['logp'].double() * wl.cuda() + out['ent'].double() * we.cuda() + out['v'].double() * wv.cuda()).sum().backward()
    torch.cuda.synchronize()
    exp, want = oracle_backward(ref, data, (wl, we, wv))
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k].detach(), exp[k].detach()) < 1e-5, (k, rel_err(out[k].detach(), exp[k].detach()))
    assert_grads(grad_report(ac.theta.grad.detach().double().cpu(), want, ac.slot_table))
    # the one-call mini-batch step (plain, and with the per-epoch weight preparation / deferred fold of ppo.train) == autograd path
    from molgym_amd import ppo as ppo_mod
    ac.theta.grad = None
    loss, info = ppo_mod.compute_loss(ac, data, 0.2, 0.5, 0.01)
    loss.backward()
    torch.cuda.synchronize()
    g_auto = ac.theta.grad.detach().clone()
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    for epoch_cache in (False, True):
        ac.theta.grad = torch.zeros_like(ac.theta)
        ac.invalidate_weights()
        stats = ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, loss_scale=0.5, epoch_cache=epoch_cache).clone()
        ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, loss_scale=0.5, epoch_cache=epoch_cache)
        ac.fold_gradients()
        torch.cuda.synchronize()
        assert (ac.theta.grad - g_auto).abs().max().item() <= 2e-5 * max(1.0, g_auto.abs().max().item()), epoch_cache
        assert abs(stats[3].item() - info['total_loss']) <= 1e-6 * max(1.0, abs(info['total_loss']))
    # rollout side: the sampling launch reads the embedded parameters too; what it reports for its draws is what evaluation gives
    ac.training = True
    torch.manual_seed(0)
    drawn = ac.step(data['obs'])
    again = ac.step(data['obs'], drawn['a'].cpu().numpy())
    assert (drawn['logp'] - again['logp']).abs().max().item() < 1e-4 and (drawn['v'] - again['v']).abs().max().item() < 1e-5
    sd = ac.state_dict()
    assert all(tuple(sd[k].shape) == tuple(v.shape) for k, v in ref.state_dict().items() if k in sd)


@pytest.mark.parametrize('levels', [2, 4])
def test_other_num_cg_levels_vs_oracle(built_lib, levels):
    """num_cg_levels = 2 / 4 (arg_parser.py:56 makes it a command-line flag; 3 is the default): a build parameter of the
    library like the channel counts (hipcc -DNLEV=..; __graft_entry__.build() pre-builds both) -- the level loops, the arena
    and the parameter layout follow it.  Outputs and every parameter gradient against the oracle built with the same
    num_cg_levels, the C-side parameter layout against molgym_amd/layout.py, and the PPO mini-batch call (forward, loss,
    backward in one call / one graph launch) against the autograd path."""
    import ctypes as C
    from molgym_amd import _lib, layout
    ac, ref, cfg = make_pair('cfg2', seed=27, num_cg_levels=levels)
    lib = ac._L()
    got = [C.c_int32() for _ in range(4)]
    lib.mg_cov_build_params(*[C.byref(g) for g in got])
    assert [g.value for g in got] == [10, 4, 4, levels] and lib is not _lib.lib()
    ccfg = ac._make_cfg(1, np.array([1]))
    n = C.c_int64()
    _lib.check(lib.mg_cov_num_params(C.byref(ccfg), C.byref(n)), lib)
    table, total = layout.offsets(len(cfg['zs']), 128, 3, 10, 4, levels)
    assert n.value == total == ac.theta.numel() and sum(p.numel() for p in ref.parameters()) == total
    assert f'cg_model.cormorant_cg.atom_levels.{levels - 1}.cat_mix.weights.0' in table
    assert f'cg_model.cormorant_cg.atom_levels.{levels}.cat_mix.weights.0' not in table
    data = make_batch(12, cfg['canvas_size'], cfg['zs'], seed=35)
    B = len(data['obs'])
    g = torch.Generator().manual_seed(3)
    wl, we, wv = (torch.randn(B, generator=g, dtype=torch.float64) * s for s in (1.0, 0.3, 0.7))
    out = ac.step(data['obs'], data['act'])
    (out['logp'].double() * wl.cuda() + out['ent'].double() * we.cuda() + out['v'].double() * wv.cuda()).sum().backward()
    torch.cuda.synchronize()
    exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    (exp['logp'] * wl + exp['ent'] * we + exp['v'] * wv).sum().backward()
    for k in ('logp', 'ent', 'v'):
        assert rel_err(out[k].detach(), exp[k].detach()) < 1e-5, (k, rel_err(out[k].detach(), exp[k].detach()))
    assert_grads(grad_report(ac.theta.grad.detach().double().cpu(), dict(ref.named_parameters()), ac.slot_table))
    # the one-call mini-batch step of ppo.train on this build
    g_auto = ac.theta.grad.detach().clone()
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    ac.theta.grad = torch.zeros_like(ac.theta)
    stats = ac.ppo_minibatch(batch, 0.2, 0.5, 0.01).clone()
    torch.cuda.synchronize()
    g_step = ac.theta.grad.detach().clone()
    ac.theta.grad = None
    from molgym_amd import ppo as ppo_mod
    loss, info = ppo_mod.compute_loss(ac, data, 0.2, 0.5, 0.01)
    loss.backward()
    torch.cuda.synchronize()
    assert (g_step - ac.theta.grad).abs().max().item() <= 2e-5 * max(1.0, g_step.abs().max().item())
    assert abs(stats[3].item() - info['total_loss']) <= 1e-6 * max(1.0, abs(info['total_loss']))
    assert torch.isfinite(stats).all() and g_auto.abs().max() > 0
