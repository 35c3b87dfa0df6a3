"""Debug tool (GPU box): host-side time of the phases of molgym_amd.ppo.train (no extra synchronisation).
usage: python tools/train_phases.py [config] [rollout] [mini batch] [epochs]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from molgym_amd import ppo  # noqa: E402
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1400
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 140
epochs = int(sys.argv[4]) if len(sys.argv) > 4 else 7
cfg = CONFIGS[name]
torch.manual_seed(0)
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'],
                 beta=cfg['beta'], device=torch.device('cuda'), **MODEL_DEFAULTS)
d = make_batch(n, cfg['canvas_size'], cfg['zs'], seed=0)
data = {k: d[k] for k in ('obs', 'act', 'logp', 'adv', 'ret')}
opt = torch.optim.Adam(ac.parameters(), lr=1e-5)
acc = {}


def timed(obj, attr, label):
    f = getattr(obj, attr)

    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, attr, w)


timed(ppo._DeviceRunner, '__init__', 'prepare_rollout')
timed(ppo._DeviceRunner, 'run', 'run (gather + launches)')
timed(ppo._DeviceRunner, 'begin_epoch', 'begin_epoch')
timed(ppo._DeviceRunner, 'end_epoch', 'end_epoch')
timed(ac, 'grad_norm_clip', 'grad_norm_clip')
timed(opt, 'step', 'optimizer.step')
timed(opt, 'zero_grad', 'zero_grad')
timed(type(ac.prepare_rollout(data)), 'minibatch', '  of which rollout.minibatch (gather)')
timed(ac, 'ppo_minibatch', '  of which ppo_minibatch')
for rep in range(3):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = ppo.train(ac, opt, data, mini_batch_size=mb, clip_ratio=0.2, target_kl=1e9, vf_coef=0.5, entropy_coef=0.01,
                     gradient_clip=0.5, max_num_steps=epochs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f'{name} rollout {n} mb {mb} epochs {epochs}: total {dt * 1e3:.1f} ms;', {k: round(v * 1e3, 2) for k, v in acc.items()},
      'unaccounted (tolist sync, stack, python)', round((dt - sum(v for k, v in acc.items() if not k.startswith('  '))) * 1e3, 2))
