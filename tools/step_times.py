"""Probe (GPU box): per-step HIP-event times of the driver's own bench schedule (`--steps 20 --warmup 5`, cadence 10) right after
start-up, and of the same 20 steps again a second later -- where the 20-step sample's +3-4 % over the long run sits."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from molgym_amd.agents.covariant import CovariantAC
from molgym_amd.spaces import ActionSpace, ObservationSpace
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch
cfg = CONFIGS['cfg2']
torch.manual_seed(0)
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'], beta=cfg['beta'], device='cuda:0', **MODEL_DEFAULTS)
d = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=0)
b = ac.prepare_batch(d['obs'], d['act'], d['logp'], d['adv'], d['ret'])
ac.theta.grad = torch.zeros_like(ac.theta)
def step(i, last):
    if i % 10 == 0:
        ac.theta.grad.zero_(); ac.invalidate_weights()
    ac.ppo_minibatch(b, 0.2, 0.5, 0.01, epoch_cache=True)
    if (i + 1) % 10 == 0 or last: ac.fold_gradients()
for i in range(5): step(i, i == 4)
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    torch.cuda.synchronize(); t0 = time.perf_counter(); ev[0].record()
    for i in range(20):
        step(i, i == 19); ev[i + 1].record()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
    print(f'rep {rep}: wall {dt / 20 * 1e3:.4f} ms/step; per step us:', ' '.join(f'{x * 1e3:.0f}' for x in per), flush=True)
    if rep == 0: time.sleep(1.0)
