"""Debug tool (GPU box): phase timestamps inside k_level0_fwd / k_level0_bwd for one workgroup.
Usage: python tools/ts_level0.py <debug .so built with -DMG_TS> [config]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from molgym_amd import _lib

_lib.LIB_PATH = sys.argv[1]
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch  # noqa: E402


def main():
    name = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']),
                     bag_scale=cfg['bag_scale'], beta=cfg['beta'], device=torch.device('cuda'), **MODEL_DEFAULTS)
    data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=0)
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    lib = _lib.lib()
    lib.mg_debug_ts.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 128)()
    for blk in (0, 100, 300, 500):  # workgroups in launch order: heavy molecules first
        lib.mg_debug_ts(buf, blk)
        for _ in range(3):
            ac.theta.grad = None
            ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
        lib.mg_debug_ts(buf, blk)
        ts = np.array(list(buf), dtype=np.int64)
        f = ts[64:70]
        b = ts[80:87]
        print(f'block {blk}: fwd phases us (a|b, c+d, e, f, g)', [round((y - x) / 100.0, 2) for x, y in zip(f[:-1], f[1:])], 'total', (f[-1] - f[0]) / 100.0)
        print(f'           bwd phases us (load+1, 2, 3, 4, 5a, 5b)', [round((y - x) / 100.0, 2) for x, y in zip(b[:-1], b[1:])], 'total', (b[-1] - b[0]) / 100.0)


if __name__ == '__main__':
    main()
