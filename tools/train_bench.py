"""Debug tool (GPU box): wall-clock throughput of molgym_amd.ppo.train (prepare_rollout + epochs of mini-batches + norm /
clip / Adam) on a synthetic rollout -- what a user of the loop sees, next to the kernel-only number of bench.py.
usage: python tools/train_bench.py [config | internal] [rollout samples] [mini batch] [epochs]"""
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
torch.manual_seed(0)
if name == 'internal':  # SchNetAC on BASELINE configs[0]
    from molgym_amd.agents.internal import SchNetAC
    from molgym_amd.synthetic import make_batch_internal
    zs = [0, 9, 16]
    ac = SchNetAC(ObservationSpace(7, zs), ActionSpace(zs), (0.8, 1.8), 128, device='cuda:0')
    d = make_batch_internal(n, 7, zs, seed=0)
else:
    cfg = CONFIGS[name]
    ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'],
                     beta=cfg['beta'], device=torch.device('cuda'), **MODEL_DEFAULTS)
    d = make_batch(n, cfg['canvas_size'], cfg['zs'], seed=0)
data = {'obs': d['obs'], 'act': d['act'], 'logp': d['logp'], 'adv': d['adv'], 'ret': d['ret']}
opt = torch.optim.Adam(ac.parameters(), lr=1e-5)
modes = [True, False, True, False] if name == 'internal' else [False, True, False, True]
for rep, mode in enumerate(modes if len(sys.argv) <= 5 else [bool(int(sys.argv[5]))] * 3):
    if mode is not None:
        ac.use_graphs = mode  # ppo_minibatch as one updated hipGraph launch (True) or ~27 stream launches (False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = ppo.train(ac, opt, data, mini_batch_size=mb, clip_ratio=0.2, target_kl=1e9, vf_coef=0.5, entropy_coef=0.01,
                     gradient_clip=0.5, max_num_steps=epochs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = info['num_opt_steps']
    print(f'{name} [graph={mode}]: rollout {n}, mini-batch {mb}: {steps} epochs in {dt * 1e3:.1f} ms -> {dt / max(steps, 1) / (n / mb) * 1e3:.3f} ms per '
          f'mini-batch, {n * steps / dt:.0f} samples/s (prepare_rollout + gathers + norm / clip / Adam included)')
