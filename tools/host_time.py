"""Debug tool (GPU box): host-side cost of one PPO mini-batch step, call by call (no device synchronisation inside).
usage: python tools/host_time.py [config]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from molgym_amd import _lib  # noqa: E402
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg = CONFIGS[name]
torch.manual_seed(0)
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'],
                 beta=cfg['beta'], device=torch.device('cuda'), **MODEL_DEFAULTS)
data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=0)
batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
ac.theta.grad = torch.zeros_like(ac.theta)
lib = _lib.lib()
acc = {}
orig = {}
for fn in ('mg_cov_forward', 'mg_ppo_loss', 'mg_cov_backward'):
    f = getattr(lib, fn)
    orig[fn] = f

    def wrap(*a, _f=f, _n=fn):
        t = time.perf_counter()
        r = _f(*a)
        acc[_n] = acc.get(_n, 0.0) + time.perf_counter() - t
        return r
    setattr(lib, fn, wrap)
for _ in range(20):
    ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
torch.cuda.synchronize()
acc.clear()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{name}: host {((t1 - t0) / n) * 1e3:.3f} ms per step (GPU-inclusive {((t2 - t0) / n) * 1e3:.3f}); inside the C calls:',
      {k: round(v / n * 1e3, 3) for k, v in acc.items()}, 'python around them:',
      round(((t1 - t0) - sum(acc.values())) / n * 1e3, 3))
