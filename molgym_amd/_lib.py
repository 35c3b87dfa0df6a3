"""ctypes binding of libmolgym_hip.so (C ABI: include/molgym_hip.h).

The product path has no CPU fallback: if the gfx950 library is missing or fails
to load, importing anything that computes raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MOLGYM_HIP_LIB') or os.path.join(_HERE, 'libmolgym_hip.so')  # override: A/B builds
MG_MAX_Z = 8


class CovCfg(C.Structure):
    _fields_ = [
        ('B', C.c_int32), ('N', C.c_int32), ('Z', C.c_int32), ('zs', C.c_int32 * MG_MAX_Z), ('W', C.c_int32),
        ('G', C.c_int32), ('TA', C.c_int32), ('TE', C.c_int32), ('has_beta', C.c_int32), ('beta', C.c_float),
        ('bag_scale', C.c_float), ('min_distance', C.c_float), ('max_distance', C.c_float),
    ]


class IntCfg(C.Structure):
    _fields_ = [
        ('B', C.c_int32), ('N', C.c_int32), ('Z', C.c_int32), ('zs', C.c_int32 * MG_MAX_Z), ('W', C.c_int32),
        ('TA', C.c_int32), ('MA', C.c_int32), ('ME', C.c_int32), ('min_distance', C.c_float),
        ('max_distance', C.c_float),
    ]


# every symbol include/molgym_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    'mg_last_error': (C.c_char_p, []),
    'mg_abi_version': (C.c_int, []),
    'mg_cov_channels': (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'mg_cov_build_params': (C.c_int, [C.POINTER(C.c_int32)] * 4),
    'mg_profile_enable': (C.c_int, [C.c_int]),
    'mg_profile_report': (C.c_int, [C.c_char_p, C.c_size_t]),
    'mg_cov_num_params': (C.c_int, [C.POINTER(CovCfg), C.POINTER(C.c_int64)]),
    'mg_cov_param_offsets': (C.c_int, [C.POINTER(CovCfg), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'mg_cov_workspace_bytes': (C.c_int, [C.POINTER(CovCfg), C.POINTER(C.c_size_t)]),
    'mg_cov_workspace_lookup': (C.c_int, [C.POINTER(CovCfg), C.c_char_p, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64)]),
    'mg_cov_forward': (C.c_int, [C.POINTER(CovCfg), _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, _P]),
    'mg_cov_sample': (C.c_int, [C.POINTER(CovCfg), _P, _P, _P, _P, _P, C.c_uint64, C.c_int32, _P, C.c_size_t, _P, _P, _P]),
    'mg_cov_sample_ids': (C.c_int, [C.POINTER(CovCfg), _P, _P, _P, _P, _P, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t,
                                    _P, _P, _P]),
    'mg_cov_backward': (C.c_int, [C.POINTER(CovCfg), _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P]),
    'mg_canvas_append': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P, _P, _P, _P, _P, _P, _P, _P]),
    'mg_adam_step': (C.c_int, [C.c_int64, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                              C.c_int64, C.c_int32, _P]),
    'mg_adam_step_gated': (C.c_int, [C.c_int64, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.c_int64, C.c_int32, _P, _P]),
    'mg_ppo_epoch_end': (C.c_int, [C.c_int64, _P, C.c_float, _P, C.c_double, C.c_double, _P, _P, _P, _P]),
    'mg_gather_rows': (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _P, C.c_int32, _P]),
    'mg_cov_check': (C.c_int, [C.POINTER(CovCfg), _P, C.c_size_t, _P]),
    'mg_cov_head_outputs': (C.c_int, [C.POINTER(CovCfg), _P, C.c_size_t, _P, _P]),
    'mg_so3_density': (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _P, _P, C.c_int32, C.c_float, _P, _P, C.c_int32, _P, _P]),
    'mg_int_num_params': (C.c_int, [C.POINTER(IntCfg), C.POINTER(C.c_int64)]),
    'mg_int_param_offsets': (C.c_int, [C.POINTER(IntCfg), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'mg_int_workspace_bytes': (C.c_int, [C.POINTER(IntCfg), C.POINTER(C.c_size_t)]),
    'mg_int_workspace_lookup': (C.c_int, [C.POINTER(IntCfg), C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'mg_int_forward': (C.c_int, [C.POINTER(IntCfg), _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, _P]),
    'mg_int_backward': (C.c_int, [C.POINTER(IntCfg), _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P]),
    'mg_cov_ppo_step': (C.c_int, [C.POINTER(CovCfg), _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P, C.c_double, C.c_double,
                                  C.c_double, C.c_double, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P]),
    'mg_cov_fold_grads': (C.c_int, [C.POINTER(CovCfg), _P, C.c_size_t, _P, _P]),
    'mg_cov_step_launches': (C.c_int, []),
    'mg_int_ppo_step': (C.c_int, [C.POINTER(IntCfg), _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P, C.c_double, C.c_double,
                                  C.c_double, C.c_double, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P]),
    'mg_ppo_loss': (C.c_int, [C.c_int32, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, _P, _P, _P]),
    'mg_gae': (C.c_int, [C.c_int32, _P, _P, _P, _P, C.c_double, C.c_double, _P, _P, _P]),
    'mg_adv_normalize': (C.c_int, [C.c_int32, _P, _P, _P]),
    'mg_grad_norm_clip': (C.c_int, [C.c_int64, _P, C.c_float, _P, _P]),
}

_lib = None
_variants = {}
DEFAULT_CHANNELS = (10, 4)  # num_channels_hidden, num_channels_per_element of the default build (arg_parser.py:55-60)
DEFAULT_LEVELS = 3  # num_cg_levels of the default build (arg_parser.py:56); 2 .. 4 are other builds of the same sources
LEVELS_RANGE = (2, 4)


def _build_key(channels):
    """(num_channels_hidden, num_channels_per_element[, num_cg_levels]) -> the full key of a library build"""
    key = tuple(int(c) for c in channels)
    return key if len(key) == 3 else key + (DEFAULT_LEVELS, )


ABI_VERSION = 9  # include/molgym_hip.h MG_ABI_VERSION: bumped whenever an entry point or the workspace layout changes
# include/molgym_hip.h MG_STEP_*: flags of mg_cov_ppo_step; mg_int_ppo_step takes WEIGHTS_CURRENT only (DEFER_FOLD: EINVAL there)
STEP_WEIGHTS_CURRENT, STEP_DEFER_FOLD = 1, 2


def _bind(path):
    # torch first: its bundled HIP runtime must be the one this library binds to (loading the library before torch
    # leaves two runtimes in the process and the second one sees no device)
    import torch  # noqa: F401
    handle = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise RuntimeError(f'{path} does not export {name}: a stale build of the library (rebuild: '
                               '`python -c "import __graft_entry__ as g; g.build()"`)') from None
        fn.restype = res
        fn.argtypes = args
    got = handle.mg_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f'{path} speaks ABI version {got}, this binding {ABI_VERSION}: a stale prebuilt library (rebuild it)')
    return handle


def variant_path(channels):
    ch, ce, nlev = _build_key(channels)
    if (ch, ce, nlev) == DEFAULT_CHANNELS + (DEFAULT_LEVELS, ):
        return LIB_PATH
    return os.path.join(_HERE, f'libmolgym_hip_c{ch}e{ce}' + ('' if nlev == DEFAULT_LEVELS else f'n{nlev}') + '.so')


def _sources_mtime():
    d = os.path.join(_HERE, 'csrc')
    srcs = [os.path.join(d, f) for f in os.listdir(d) if f.endswith(('.hip', '.inc', '.h'))]
    header = os.path.join(os.path.dirname(_HERE), 'include', 'molgym_hip.h')  # MG_ABI_VERSION lives there
    if os.path.exists(header):
        srcs.append(header)
    return max(os.path.getmtime(f) for f in srcs)


def build_variant(channels, force=False):
    """hipcc build of the same sources for other channel counts / another num_cg_levels (compile-time constants of the
    kernels; `channels` = (C, Ce) or (C, Ce, num_cg_levels)): in-tree, next to the default library.  Needs hipcc (present in the ROCm image on both the build container and the GPU box).
    Several ranks of one torchrun launch may get here at once: the build is serialised by a file lock, goes to a temporary
    file in the same directory and is moved into place atomically, so nobody ever dlopens a half-written library."""
    import fcntl
    import subprocess
    import tempfile
    channels = _build_key(channels)
    path = variant_path(channels)
    src = os.path.join(_HERE, 'csrc', 'molgym_hip.hip')

    def fresh():
        return os.path.exists(path) and os.path.getmtime(path) >= _sources_mtime()

    if not force and fresh():
        return path
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        raise RuntimeError(f'num_channels_hidden={channels[0]}, num_channels_per_element={channels[1]}, num_cg_levels={channels[2]} need their own build '
                           f'of the library and {hipcc} is not there')
    try:
        lock = open(path + '.lock', 'w')
    except OSError as e:
        raise RuntimeError(f'cannot build {path}: the package directory is not writable ({e}); build the variant once where '
                           'it is (`python -c "from molgym_amd import _lib; _lib.build_variant((C, Ce))"`)') from None
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():  # another process built it while this one waited for the lock
                return path
            fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + '.', suffix='.tmp', dir=_HERE)
            os.close(fd)
            try:
                subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                                       f'-DCH={channels[0]}', f'-DCE={channels[1]}', f'-DNLEV={channels[2]}', src, '-o', tmp])
                os.chmod(tmp, 0o755)
                os.replace(tmp, path)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return path


MAX_CHANNELS = (10, 5)  # csrc/common.h static_asserts: NLM * CH <= 256, 2 * NLM * CE <= 256


def lib(channels=None):
    """Load (once) and return the library; raises if it is not built.  `channels` = (num_channels_hidden,
    num_channels_per_element[, num_cg_levels]) selects the build of the Cormorant kernels (None / (10, 4) / (10, 4, 3): the
    default library; anything else is compiled on first use, see build_variant)."""
    global _lib
    if channels is not None and _build_key(channels) != DEFAULT_CHANNELS + (DEFAULT_LEVELS, ):
        key = _build_key(channels)
        if key not in _variants:
            if not LEVELS_RANGE[0] <= key[2] <= LEVELS_RANGE[1]:
                raise RuntimeError(f'num_cg_levels {key[2]} outside what the kernels were written for '
                                   f'({LEVELS_RANGE[0]}..{LEVELS_RANGE[1]})')
            if key[0] < 1 or key[0] > MAX_CHANNELS[0] or key[1] < 1 or key[1] > MAX_CHANNELS[1]:
                raise RuntimeError(f'num_channels_hidden {key[0]} / num_channels_per_element {key[1]} outside what the '
                                   f'kernels were written for (1..{MAX_CHANNELS[0]} / 1..{MAX_CHANNELS[1]}: one 256-thread '
                                   'workgroup covers the 25 * C items of an atom)')
            handle = _bind(build_variant(key))
            ch, ce, ml, nl = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            handle.mg_cov_build_params(C.byref(ch), C.byref(ce), C.byref(ml), C.byref(nl))
            if (ch.value, ce.value, nl.value) != key:
                raise RuntimeError(f'{variant_path(key)} was built for channels / levels {(ch.value, ce.value, nl.value)}')
            _variants[key] = handle
        return _variants[key]
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
        _lib = _bind(LIB_PATH)
    return _lib


def check(rc, handle=None):
    if rc != 0:
        raise RuntimeError(f'molgym_hip error {rc}: {(handle or lib()).mg_last_error().decode()}')


def gather_rows(tensors, idx_dev, stream_ptr):
    """rows `idx_dev` (int64, on the device) of up to 8 row-major device tensors in ONE launch (mg_gather_rows); the results
    are views of one allocation, each contiguous.  Replaces one index_select per tensor (host time, not device time, is
    what those cost on a 140-sample mini-batch)."""
    import torch
    B = int(idx_dev.numel())
    nf = len(tensors)
    rb = [int(t[0].numel() * t.element_size()) if t.dim() > 1 else int(t.element_size()) for t in tensors]
    offs, total = [], 0
    for r in rb:
        offs.append(total)
        total += (B * r + 63) // 64 * 64
    buf = torch.empty(max(total, 64), dtype=torch.uint8, device=idx_dev.device)
    outs = []
    for t, r, o in zip(tensors, rb, offs):
        outs.append(buf[o:o + B * r].view(t.dtype).view((B, ) + tuple(t.shape[1:])))
    if B == 0:
        return outs
    src = (C.c_void_p * nf)(*[t.data_ptr() for t in tensors])
    dst = (C.c_void_p * nf)(*[buf.data_ptr() + o for o in offs])
    rbs = (C.c_int32 * nf)(*rb)
    check(lib().mg_gather_rows(nf, src, dst, rbs, C.c_void_p(idx_dev.data_ptr()), B, stream_ptr))
    return outs

