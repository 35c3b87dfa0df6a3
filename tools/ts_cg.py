"""Debug tool (GPU box): phase timestamps (100 MHz ticks) inside k_catbuild_bwd_mfma for one wave.
Usage: python tools/ts_cg.py <debug .so built with -DMG_TS> [config]"""
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

name = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
cfg = CONFIGS[name]
torch.manual_seed(0)
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'],
                 beta=cfg['beta'], device=torch.device('cuda'), **MODEL_DEFAULTS)
data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=0)
batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
lib = _lib.lib()
lib.mg_debug_ts.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 128)()
for blk in (0, 8, 801):
    lib.mg_debug_ts(buf, blk)
    for _ in range(3):
        ac.theta.grad = None
        ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
    lib.mg_debug_ts(buf, blk)
    ts = np.array(list(buf), dtype=np.int64)
    us = lambda a, b: round((ts[b] - ts[a]) / 100.0, 2)
    print('block', blk, 'bwd: issue loads', us(32, 54), '| wait + LDS stores', us(54, 55), '| barrier', us(55, 33), '| rebuild power', us(33, 50),
          '| aD + mfma column', us(50, 51), '| atomics + sync', us(51, 34), '| rebuild aggregate', us(34, 52), '| aD + tile store', us(52, 35),
          '| tile 0: operands + mfma', us(35, 36), '| P1 epilogue', us(36, 53), '| P2 epilogue', us(53, 37), '| dE', us(37, 38),
          '| other tiles', us(38, 39), '| total', us(32, 39))
    print('   fwd: loads', us(40, 41), '| mfma', us(41, 42), '| projection', us(42, 43))
