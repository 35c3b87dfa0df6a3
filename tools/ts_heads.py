"""Debug tool (GPU box): phase timestamps inside k_heads_fwd / k_heads_bwd for one workgroup.
Usage: python tools/ts_heads.py <debug .so built with -DMG_TS> [config]"""
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
    natoms = [sum(1 for it in o[0] if cfg['zs'][it[0]] != 0) for o in data['obs']]
    lib = _lib.lib()
    lib.mg_debug_ts.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 128)()
    for target in (max(natoms), int(np.median(natoms)), -max(natoms)):
        blk = natoms.index(abs(target))
        if target < 0:  # stamp the first ATOM of that sample (k_catbuild_bwd is one workgroup per atom)
            blk = int(sum(natoms[:blk]))
        lib.mg_debug_ts(buf, blk)
        for _ in range(3):
            ac.theta.grad = None
            ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
        lib.mg_debug_ts(buf, blk)
        ts = np.array(list(buf), dtype=np.int64)
        print(f'block {blk} natoms {target}')
        print('  catbuild_bwd phases (stage, rebuild, tile-stage, tile-compute, tile-reduce, rest) us', [round((ts[j] - ts[i]) / 100.0, 1) for i, j in ((32, 33), (33, 34), (34, 35), (35, 36), (36, 37), (37, 38))], 'total', (ts[38] - ts[32]) / 100.0)
        print('  bwd 5->11->12->6 us', [round((ts[j] - ts[i]) / 100.0, 1) for i, j in ((5, 11), (11, 12), (12, 6))])
        print('  bwd 2->13->28->3 us', [round((ts[j] - ts[i]) / 100.0, 1) for i, j in ((2, 13), (13, 28), (28, 3))])
        print('  bwd 9->14->15->10 us', [round((ts[j] - ts[i]) / 100.0, 1) for i, j in ((9, 14), (14, 15), (15, 10))])
        print('  fwd us since kernel start: role0 (16) ->17,18,19,44', [round((ts[j] - ts[16]) / 100.0, 1) for j in (17, 18, 19, 44)], '| role1 45,20..23', [round((ts[j] - ts[16]) / 100.0, 1) for j in (45, 20, 21, 22, 23)], '| role2 49,24..27', [round((ts[j] - ts[16]) / 100.0, 1) for j in (49, 24, 25, 26, 27)])
        print('  fwd role2 start: 49 -> 61 (loads, staging) -> 62 (barrier) -> 63 (cat rows, CG power) -> 24 us', [round((ts[j] - ts[i]) / 100.0, 1) for i, j in ((49, 61), (61, 62), (62, 63), (63, 24))])
        print('  fwd role2 so3: 26 -> 56 (setup) -> 57 (pass 0) -> 58 (passes) -> 27 (reductions) us', [round((ts[j] - ts[i]) / 100.0, 1) for i, j in ((26, 56), (56, 57), (57, 58), (58, 27))])
        print('  bwd role1 47,28,3,4,5,11,12,6,7,8,48 us since 0:', [round((ts[j] - ts[0]) / 100.0, 1) for j in (47, 28, 3, 4, 5, 11, 12, 6, 7, 8, 48)])
        print('  bwd role1 start: 47 -> 59 (staging + barrier) -> 60 (thread 0: drawn orientation) -> 28 (Lebedev passes) us', [round((ts[j] - ts[i]) / 100.0, 1) for i, j in ((47, 59), (59, 60), (60, 28))])
        for nm, lo, hi in (('bwd', 0, 10),):
            seg = ts[lo:hi + 1]
            print(' ', nm, 'total us', (seg[-1] - seg[0]) / 100.0, 'phases us', [round((b - a) / 100.0, 1) for a, b in zip(seg[:-1], seg[1:])])


if __name__ == '__main__':
    main()
