"""Host-side logic that needs no GPU: observation parsing, parameter layout, the C ABI surface."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from molgym_amd import layout
from molgym_amd.agents.covariant import parse_observations_host
from molgym_amd.lebedev import lebedev_table
from molgym_amd.synthetic import CONFIGS, make_batch
from oracle.covariant_ref import parse_observations


def test_parse_matches_oracle_and_compacts_null_items():
    cfg = CONFIGS['cfg2']
    data = make_batch(9, cfg['canvas_size'], cfg['zs'], seed=2)
    obs = list(data['obs'])
    # a null item in the middle of a canvas must be dropped and the rest compacted (spaces.py:55-61)
    c, bag = obs[1]
    c = list(c)
    c[2] = (0, (0.0, 0.0, 0.0))
    obs[1] = (tuple(c), bag)
    pos, charges, bags, natoms = parse_observations_host(obs, cfg['zs'], cfg['canvas_size'])
    ref = parse_observations(obs, cfg['zs'], cfg['canvas_size'])
    np.testing.assert_array_equal(charges, ref['charges'].numpy())
    np.testing.assert_allclose(pos, ref['positions'].numpy())
    np.testing.assert_array_equal(natoms, ref['num_atoms'].numpy())
    np.testing.assert_array_equal(bags, ref['bags'].numpy())
    assert natoms[0] == 0 and natoms[1] == cfg['canvas_size'] - 1


def test_parse_rejects_malformed_input():
    cfg = CONFIGS['cfg2']
    data = make_batch(2, cfg['canvas_size'], cfg['zs'], seed=0)
    with pytest.raises(RuntimeError):
        parse_observations_host([(data['obs'][0][0][:3], data['obs'][0][1])], cfg['zs'], cfg['canvas_size'])
    with pytest.raises(RuntimeError):
        bad = ((99, (0., 0., 0.)), ) * cfg['canvas_size']
        parse_observations_host([(bad, data['obs'][0][1])], cfg['zs'], cfg['canvas_size'])


def test_layout_matches_library_and_reference_counts(built_lib):
    from molgym_amd import _lib
    for zs, width, total in (([0, 9, 16], 128, 185006), ([0, 1, 6, 7, 8], 128, None), ([0, 1, 6, 8], 64, None)):
        table, n = layout.offsets(len(zs), width, 3)
        cfg = _lib.CovCfg()
        cfg.B, cfg.N, cfg.Z, cfg.W, cfg.G, cfg.TA, cfg.TE = 2, 7, len(zs), width, 3, 3, 5
        for i, z in enumerate(zs):
            cfg.zs[i] = z
        cfg.min_distance, cfg.max_distance, cfg.bag_scale = 0.8, 1.8, 5
        cnt = C.c_int64()
        _lib.check(built_lib.mg_cov_num_params(C.byref(cfg), C.byref(cnt)))
        assert cnt.value == n and (total is None or n == total)
        offs = (C.c_int64 * 256)()
        ns = C.c_int32()
        _lib.check(built_lib.mg_cov_param_offsets(C.byref(cfg), offs, C.byref(ns)))
        assert ns.value == len(table)
        assert [offs[i] for i in range(ns.value)] == [o for o, _ in table.values()]


def test_library_exports_every_declared_symbol(built_lib):
    import os
    from molgym_amd import _lib
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include',
                               'molgym_hip.h')).read()
    declared = set(re.findall(r'\b(mg_[a-z0-9_]+)\s*\(', header)) - {'mg_cov_cfg'}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(built_lib, name)
    # one version number in three places: the header's define, what the library returns, what the binding accepts
    header_version = int(re.search(r'#define\s+MG_ABI_VERSION\s+(\d+)', header).group(1))
    assert built_lib.mg_abi_version() == header_version == _lib.ABI_VERSION


def test_channel_counts_outside_the_kernels_range_are_refused():
    """the kernels cover the 25 * C items of an atom with one 256-thread workgroup (static_asserts in csrc/common.h): the
    binding refuses the builds that could not compile instead of starting hipcc"""
    from molgym_amd import _lib
    for bad in ((11, 4), (16, 4), (10, 6), (0, 4)):
        with pytest.raises(RuntimeError):
            _lib.lib(bad)


def test_num_cg_levels_is_a_build_parameter_with_its_own_library_and_layout(built_lib):
    """num_cg_levels (arg_parser.py:56) selects a build of the library like the channel counts: its own file name, its own
    parameter layout (molgym_amd/layout.py and the C side agree slot by slot); values outside 2..4 are refused before hipcc
    starts.  __graft_entry__.build() pre-builds 2 and 4; the parity tests against the oracle are `-m gpu`."""
    import os
    from molgym_amd import _lib, layout
    assert _lib.variant_path((10, 4)) == _lib.variant_path((10, 4, 3)) == _lib.LIB_PATH
    assert _lib.variant_path((10, 4, 2)).endswith('libmolgym_hip_c10e4n2.so')
    assert _lib.variant_path((8, 2)).endswith('libmolgym_hip_c8e2.so') and _lib.variant_path((8, 2, 4)).endswith('libmolgym_hip_c8e2n4.so')
    for bad in ((10, 4, 1), (10, 4, 5), (10, 4, 0)):
        with pytest.raises(RuntimeError):
            _lib.lib(bad)
    for levels in (2, 4):
        if not os.path.exists(_lib.variant_path((10, 4, levels))):
            pytest.skip('variant libraries not built (python -c "import __graft_entry__ as g; g.build()")')
        lib = _lib.lib((10, 4, levels))
        got = [C.c_int32() for _ in range(4)]
        lib.mg_cov_build_params(*[C.byref(g) for g in got])
        assert [g.value for g in got] == [10, 4, 4, levels]
        cfg = _lib.CovCfg()
        cfg.B, cfg.N, cfg.Z, cfg.W, cfg.G, cfg.TA, cfg.TE = 2, 7, 3, 128, 3, 3, 5
        cfg.zs[0], cfg.zs[1], cfg.zs[2] = 0, 9, 16
        cfg.min_distance, cfg.max_distance, cfg.bag_scale = 0.8, 1.8, 5
        n = C.c_int64()
        _lib.check(lib.mg_cov_num_params(C.byref(cfg), C.byref(n)), lib)
        table, total = layout.offsets(3, 128, 3, 10, 4, levels)
        assert n.value == total
        offs = (C.c_int64 * 256)()
        ns = C.c_int32()
        _lib.check(lib.mg_cov_param_offsets(C.byref(cfg), offs, C.byref(ns)), lib)
        assert ns.value == len(table) and [offs[i] for i in range(ns.value)] == [o for o, _ in table.values()]


def test_invalid_configuration_is_reported(built_lib):
    from molgym_amd import _lib
    cfg = _lib.CovCfg()
    cfg.B, cfg.N, cfg.Z, cfg.W, cfg.G = 1, 7, 1, 128, 3
    n = C.c_int64()
    assert built_lib.mg_cov_num_params(C.byref(cfg), C.byref(n)) < 0
    assert b'Z=' in built_lib.mg_last_error()


def test_lebedev_table_integrates_harmonics_exactly():
    tab = lebedev_table().astype(np.float64)
    w = np.exp(tab[50])
    assert abs(w.sum() - 1) < 1e-6
    y = tab[0:50:2] + 1j * tab[1:50:2]  # (25, 1730)
    gram = (y * w) @ y.conj().T * 4 * np.pi  # orthonormality of Y_lm under the quadrature
    np.testing.assert_allclose(gram, np.eye(25), atol=2e-6)


def test_lebedev_table_has_the_symmetries_the_heads_kernels_read_it_with():
    """[r6] k_heads_fwd / k_heads_bwd fetch 26 of the table's 51 rows (heads_fused.inc: leb_fetch26 / leb_y) and take the m < 0
    rows from the m > 0 rows: Y_l,-m = (-1)^m conj(Y_l,m) and Im Y_l,0 = 0 must hold EXACTLY in the float32 table, and point
    g + NLEB/2 must be exactly -point g with the same weight (the antipodal pairing both kernels use)"""
    tab = lebedev_table()
    assert tab.dtype == np.float32 and tab.shape == (51, 1730)
    for l in range(5):
        q0 = l * l + l
        assert np.all(tab[2 * q0 + 1] == 0.0)
        for m in range(1, l + 1):
            sg = np.float32(-1.0 if m & 1 else 1.0)
            qp, qm = q0 + m, q0 - m
            np.testing.assert_array_equal(tab[2 * qm], sg * tab[2 * qp])
            np.testing.assert_array_equal(tab[2 * qm + 1], -sg * tab[2 * qp + 1])
    h = 1730 // 2
    np.testing.assert_array_equal(tab[50, :h], tab[50, h:])
    for q in range(25):  # Y_lm(-x) = (-1)^l Y_lm(x), exactly
        l = int(np.floor(np.sqrt(q)))
        sg = np.float32(-1.0 if l & 1 else 1.0)
        np.testing.assert_array_equal(tab[2 * q, h:], sg * tab[2 * q, :h])
        np.testing.assert_array_equal(tab[2 * q + 1, h:], sg * tab[2 * q + 1, :h])


def test_product_path_never_imports_the_oracle():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, 'molgym_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_internal_batched_action_conversion_equals_per_sample():
    """SchNetAC's rollout step converts the drawn action rows with ONE vectorised z-matrix placement
    (`_actions_to_space`); it must give exactly what `to_action_space` (internal/agent.py:91-110) gives row by row,
    including the kappa sign flip and canvases of 0, 1 and 2 atoms (the fixed-axis special cases of zmat.py)."""
    import numpy as np
    from molgym_amd.agents.internal import SchNetAC
    from molgym_amd.spaces import ActionSpace, ObservationSpace
    from molgym_amd.synthetic import CONFIGS, make_batch
    cfg = CONFIGS['cfg2']
    ia = SchNetAC.__new__(SchNetAC)  # host-side helpers only: no device, no parameters
    ia.observation_space, ia.action_space = ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs'])
    ia.zs, ia.num_atoms, ia.num_zs = list(cfg['zs']), cfg['canvas_size'], len(cfg['zs'])
    data = make_batch(80, cfg['canvas_size'], cfg['zs'], seed=1)
    rng = np.random.default_rng(0)
    _, _, natoms, pos64 = ia._parse(data['obs'])
    assert {0, 1, 2} <= set(natoms.tolist())
    acts = np.zeros((80, 7), dtype=np.float32)
    acts[:, 1] = [rng.integers(0, max(n, 1)) for n in natoms]
    acts[:, 2] = rng.integers(1, 3, 80)
    acts[:, 3], acts[:, 4], acts[:, 5] = rng.uniform(0.8, 1.8, 80), rng.uniform(0.2, 2.9, 80), rng.uniform(0.1, 3.0, 80)
    acts[:, 6] = rng.integers(0, 2, 80)
    got = ia._actions_to_space(acts, pos64, natoms)
    want = [ia.to_action_space(a, o) for a, o in zip(acts, data['obs'])]
    for g, w in zip(got, want):
        assert g[0] == w[0] and np.array_equal(np.asarray(g[1]), np.asarray(w[1]), equal_nan=True)


def test_native_observation_parser_equals_the_numpy_path():
    """molgym_amd/csrc/obsparse.c (one C traversal of the observation tuples, built by __graft_entry__.build()) fills the same
    three arrays as the list comprehensions + np.array of ParsedObservations.from_list -- tuples, lists and numpy scalars alike --
    and irregular input still ends in the numpy path's RuntimeError."""
    from molgym_amd import observations as om
    native = om._native_parser()
    if native is None:
        pytest.skip('molgym_amd/_obsparse.so not built (python -c "import __graft_entry__ as g; g.build()")')
    cfg = CONFIGS['cfg2']
    data = make_batch(37, cfg['canvas_size'], cfg['zs'], seed=5)
    obs = list(data['obs'])
    obs[3] = ([[np.int64(l), [np.float64(c) for c in p]] for l, p in obs[3][0]], list(obs[3][1]))  # lists + numpy scalars
    got = om.ParsedObservations.from_list(obs)
    labels = np.array([[item[0] for item in o[0]] for o in obs], dtype=np.int64)
    xyz = np.array([[item[1] for item in o[0]] for o in obs], dtype=np.float64)
    bags = np.array([o[1] for o in obs], dtype=np.int64)
    assert np.array_equal(got.labels, labels) and np.array_equal(got.xyz, xyz) and np.array_equal(got.bags, bags)
    saved, om._NATIVE = om._NATIVE, None  # the numpy path on the same input
    try:
        ref = om.ParsedObservations.from_list(obs)
    finally:
        om._NATIVE = saved
    assert np.array_equal(ref.labels, got.labels) and np.array_equal(ref.xyz, got.xyz) and np.array_equal(ref.bags, got.bags)
    with pytest.raises(RuntimeError):
        om.ParsedObservations.from_list([obs[0], (obs[1][0][:3], obs[1][1])])


def test_native_observation_parser_under_address_sanitizer(tmp_path):
    """SURVEY section 5's sanitizer pass for the one piece of native HOST code on the path: csrc/obsparse.c rebuilt with
    -fsanitize=address,undefined and driven in a child interpreter (libasan preloaded) over regular rollouts, tuples / lists /
    numpy scalars, empty input and the irregular inputs that must end in an exception -- any out-of-bounds access, use after
    free or reference-count slip on those paths aborts the child."""
    import shutil
    import subprocess
    import sys
    import sysconfig
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    libasan = subprocess.run([gcc, '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip('no libasan')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = tmp_path / 'asanpkg'
    pkg.mkdir()
    ext = pkg / ('_obsparse' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))
    subprocess.check_call([gcc, '-O1', '-g', '-fno-omit-frame-pointer', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                           '-shared', '-fPIC', '-I' + sysconfig.get_paths()['include'],
                           os.path.join(root, 'molgym_amd', 'csrc', 'obsparse.c'), '-o', str(ext)])
    script = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import _obsparse
assert _obsparse.__file__.startswith(sys.argv[2]), _obsparse.__file__
from molgym_amd import observations as om
from molgym_amd.synthetic import CONFIGS, make_batch
om._NATIVE = _obsparse
for name, n in (('cfg2', 140), ('cfg3', 33), ('cfg5', 9)):
    cfg = CONFIGS[name]
    obs = list(make_batch(n, cfg['canvas_size'], cfg['zs'], seed=n)['obs'])
    obs[1] = ([[np.int64(l), [np.float64(c) for c in p]] for l, p in obs[1][0]], list(obs[1][1]))
    got = om.ParsedObservations.from_list(obs)
    om._NATIVE = None
    ref = om.ParsedObservations.from_list(obs)
    om._NATIVE = _obsparse
    assert np.array_equal(got.labels, ref.labels) and np.array_equal(got.xyz, ref.xyz) and np.array_equal(got.bags, ref.bags)
    bad = [[obs[0], (obs[1][0][:3], obs[1][1])], [obs[0], (obs[1][0], obs[1][1][:1])], [obs[0], None], [obs[0], (None, None)],
           [obs[0], ([(0, (0.0, 0.0))] * len(obs[0][0]), obs[0][1])], [obs[0], ([('x', (0.0, 0.0, 0.0))] * len(obs[0][0]), obs[0][1])]]
    for b in bad:
        try:
            om.ParsedObservations.from_list(b)
        except Exception:
            pass
        else:
            raise SystemExit('irregular input was accepted')
print('asan clean')
'''
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', PYTHONMALLOC='malloc')
    r = subprocess.run([sys.executable, '-c', script, root, str(pkg)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and 'asan clean' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize('maxl', [3, 2, 1])
def test_maxl_embedding_is_exact_on_the_oracle(maxl):
    """[r6] --maxl below 4 (arg_parser.py:56): CovariantAC keeps the maxl-limited parameter vector (the reference's state_dict
    shapes) and hands the kernels its EMBEDDING into the maxl = 4 layout (layout.embedding_index).  The claim that makes this
    exact -- the smaller network is the larger one with every quantity of a degree above maxl multiplied by a zero weight -- is
    checked here on the oracle alone (float64, CPU): outputs and the gathered gradient of the embedded maxl = 4 oracle equal
    those of the maxl-limited oracle; and the parameter layout equals the maxl-limited oracle module's state_dict."""
    import numpy as np
    import torch
    from molgym_amd import layout
    from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch
    from oracle.covariant_ref import CovariantACRef
    cfg = CONFIGS['cfg2']
    Z = len(cfg['zs'])
    data = make_batch(5, cfg['canvas_size'], cfg['zs'], seed=3)
    torch.manual_seed(maxl)
    kw = dict(MODEL_DEFAULTS, maxl=maxl)
    common = dict(zs=cfg['zs'], canvas_size=cfg['canvas_size'], bag_scale=cfg['bag_scale'], beta=cfg['beta'])
    small = CovariantACRef(**common, **kw).double()
    big = CovariantACRef(**common, **MODEL_DEFAULTS).double()
    ts, ns = layout.offsets(Z, 128, 3, 10, 4, 3, maxl)
    tb, nb = layout.offsets(Z, 128, 3, 10, 4, 3, 4)
    sd = small.state_dict()
    assert ns == sum(p.numel() for p in small.parameters()) and all(tuple(sd[k].shape) == tuple(s) for k, (_, s) in ts.items())
    idx = torch.from_numpy(layout.embedding_index(Z, 128, 3, 10, 4, 3, maxl))
    assert len(idx) == ns and len(torch.unique(idx)) == ns and int(idx.max()) < nb
    ps, pb = dict(small.named_parameters()), dict(big.named_parameters())
    flat = torch.zeros(nb, dtype=torch.float64)
    flat[idx] = torch.cat([ps[k].detach().reshape(-1) for k in ts])
    big.load_state_dict({k: flat[o:o + int(np.prod(s))].view(s).clone() for k, (o, s) in tb.items()}, strict=False)
    w = torch.randn(5, dtype=torch.float64)
    grads = []
    for ref, table, params in ((small, ts, ps), (big, tb, pb)):
        out = ref.step(data['obs'], data['act'], dtype=torch.float64)
        (out['logp'] * w + out['v'] + 0.1 * out['ent']).sum().backward()
        grads.append((out, torch.cat([(params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])).reshape(-1)
                                      for k in table])))
    (os_, gs), (ob, gb) = grads
    for k in ('logp', 'ent', 'v'):
        assert (os_[k] - ob[k]).abs().max().item() <= 1e-12, k
    assert (gs - gb[idx]).abs().max().item() <= 1e-12 * max(1.0, gs.abs().max().item())
