"""Debug tool (GPU box): does one PPO mini-batch step (forward + loss + backward) capture into a HIP graph, and what does a
replay cost next to the eager launches?  usage: python tools/graph_probe.py [config]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg = CONFIGS[name]
torch.manual_seed(0)
dev = torch.device('cuda')
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'],
                 beta=cfg['beta'], device=dev, **MODEL_DEFAULTS)
data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=0)
batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
ac.theta.grad = torch.zeros_like(ac.theta)


def step():
    ac.theta.grad.zero_()
    return ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3


eager = timeit(step)
ref_stats = step().clone()
ref_grad = ac.theta.grad.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
g.replay()
torch.cuda.synchronize()
print('stats equal', torch.allclose(out, ref_stats, rtol=1e-6), 'grad rel err',
      ((ac.theta.grad - ref_grad).abs().max() / ref_grad.abs().max()).item())
graphed = timeit(g.replay)
print(f'{name}: eager {eager[0]:.4f} ms/step (host enqueue {eager[1]:.4f}), graph replay {graphed[0]:.4f} ms/step (host {graphed[1]:.4f})')
