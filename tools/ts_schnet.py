"""Debug tool (GPU box): phase timestamps inside k_schnet_fwd for one molecule's workgroup.
Usage: python tools/ts_schnet.py <debug .so built with -DMG_TS>"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from molgym_amd import _lib

_lib.LIB_PATH = sys.argv[1]
from molgym_amd.agents.internal import SchNetAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import make_batch_internal  # noqa: E402

ZS, N = [0, 9, 16], 7
ac = SchNetAC(ObservationSpace(N, ZS), ActionSpace(ZS), (0.8, 1.8), 128, device='cuda:0')
data = make_batch_internal(140, N, ZS, seed=0)
lib = _lib.lib()
lib.mg_debug_ts.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 128)()
for blk in (0, 5, 200, 400):
    lib.mg_debug_ts(buf, blk)
    for _ in range(3):
        out = ac.step(data['obs'], data['act'])
    torch.cuda.synchronize()
    lib.mg_debug_ts(buf, blk)
    ts = np.array(list(buf), dtype=np.int64) / 100.0
    print('molecule', blk, 'stage %.1f' % (ts[99] - ts[98]))
    for it in range(3):
        b = 100 + 8 * it
        prev = ts[99] if it == 0 else ts[b - 2]
        print('   t=%d: in2f %.1f | cfconv %.1f | store f2o %.1f | f2out+ssp %.1f | store dense %.1f | dense+resid %.1f | store next %.1f'
              % (it, ts[b] - prev, ts[b + 1] - ts[b], ts[b + 2] - ts[b + 1], ts[b + 3] - ts[b + 2], ts[b + 4] - ts[b + 3],
                 ts[b + 5] - ts[b + 4], ts[b + 6] - ts[b + 5]))
    print('   total %.1f us' % (ts[122] - ts[98]))
