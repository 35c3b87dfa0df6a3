"""Debug tool (GPU box): forward start / end and backward start / end of EVERY workgroup of the two heads kernels (one
mini-batch step): which role of which sample the kernels wait for.
Usage: python tools/ts_heads_wg.py <debug .so built with -DMG_TS> [config]"""
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
    natoms = np.array([sum(1 for it in o[0] if cfg['zs'][it[0]] != 0) for o in data['obs']])
    B = len(natoms)
    lib = _lib.lib()
    lib.mg_debug_wg_ts.argtypes = [C.c_void_p]
    buf = (C.c_ulonglong * (3 * 256 * 4))()
    for _ in range(4):
        ac.theta.grad = None
        ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
    lib.mg_debug_wg_ts(buf)
    ts = np.array(list(buf), dtype=np.int64).reshape(3, 256, 4)[:, :B] / 100.0  # us
    t0 = ts[:, :, 0].min()
    tb = ts[:, :, 2].min()  # (separate kernels: the backward's first start)
    print('forward roles: 0 = per-atom MLPs / focus / value, 1 = element + distance, 2 = mixer + orientation')
    q = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 50, 90, 100])]
    for r in range(3):
        print(f'role {r}: start since first start (min/med/p90/max) {q(ts[r, :, 0] - t0)}  forward {q(ts[r, :, 1] - ts[r, :, 0])}  '
              f'forward end {q(ts[r, :, 1] - t0)}')
        print(f'         backward start - forward end {q(ts[r, :, 2] - ts[r, :, 1])}  backward {q(ts[r, :, 3] - ts[r, :, 2])}  '
              f'backward start since first {q(ts[r, :, 2] - tb)}  end since first start {q(ts[r, :, 3] - t0)}')
        for n in (0, 3, 7):
            m = natoms == n
            if m.any():
                print(f'         n = {n}: forward med {np.median(ts[r, m, 1] - ts[r, m, 0]):.1f}  backward med '
                      f'{np.median(ts[r, m, 3] - ts[r, m, 2]):.1f}')
    print('whole: first start -> last end', round(float(ts[:, :, 3].max() - t0), 1), 'us; backward window',
          round(float(ts[:, :, 3].max() - tb), 1))


if __name__ == '__main__':
    main()
