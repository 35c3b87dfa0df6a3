"""Debug tool (GPU box): phase timestamps inside k_int_heads_fwd_pre / k_int_heads_bwd_pre for one sample's workgroup.
Usage: python tools/ts_int_heads.py <debug .so built with -DMG_TS>"""
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
FWD = [(64, 'start'), (65, 'loads issued'), (66, 'rows in LDS'), (67, 'beta0'), (68, 'beta1'), (69, 'input rows'), (75, 'first layers: partials'),
       (70, 'barrier'), (71, 'sums'), (72, 'last layers'), (73, 'hV2'), (74, 'final wave')]
BWD = [(80, 'start'), (81, 'loads issued'), (82, 'head adjoints'), (83, 'barrier'), (84, 'hidden adjoints'), (85, 'input-row partials'),
       (86, 'barrier'), (87, 'sums'), (88, 'critic partial'), (89, 'dx3 / d_lbag'), (90, 'beta partial'), (91, 'end')]
for blk in (0, 5, 70, 139):
    lib.mg_debug_ts(buf, blk)
    for _ in range(3):
        out = ac.step(data['obs'], data['act'])
        (out['logp'].sum() + out['v'].sum() + out['ent'].sum()).backward()
    torch.cuda.synchronize()
    lib.mg_debug_ts(buf, blk)
    ts = np.array(list(buf), dtype=np.int64) / 100.0
    for nm, marks in (('fwd', FWD), ('bwd', BWD)):
        print('sample', blk, nm, ' | '.join('%s %.1f' % (lab, ts[i] - ts[marks[k - 1][0]]) for k, (i, lab) in enumerate(marks) if k),
              '| total %.1f us' % (ts[marks[-1][0]] - ts[marks[0][0]]))
