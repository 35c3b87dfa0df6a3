"""Rollout-side step(obs) timing (GPU box): samples/s of the sampling path for both agents.
usage: python tools/rollout_bench.py [config] [batch]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.agents.internal import SchNetAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    cfg = CONFIGS[name]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    obs = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=0)['obs']
    osp, asp = ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs'])
    torch.manual_seed(0)
    ac = CovariantAC(osp, asp, bag_scale=cfg['bag_scale'], beta=cfg['beta'], device=torch.device('cuda:0'), **MODEL_DEFAULTS)
    ac.training = True
    dt = timeit(lambda: ac.step(obs))
    print(f'covariant step(obs) B={B}: {dt * 1e3:.3f} ms -> {B / dt:.0f} samples/s (host parse + D2H of the actions included)')
    canvas = ac.make_canvas(obs)
    dt = timeit(lambda: ac.step_canvas(canvas, commit=False))
    print(f'covariant step_canvas B={B}: {dt * 1e3:.3f} ms -> {B / dt:.0f} samples/s (device-resident canvases: no parse; '
          f'D2H of the actions and placed positions included)')
    t0 = time.perf_counter()
    for _ in range(20):
        canvas.sync(list(range(0, B, 7)), [obs[i] for i in range(0, B, 7)])
    torch.cuda.synchronize()
    print(f'  canvas.sync of {len(range(0, B, 7))} reset environments: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms')
    ia = SchNetAC(osp, asp, (0.8, 1.8), 128, device='cuda:0')
    ia.training = True
    dt = timeit(lambda: ia.step(obs), n=5, warm=1)
    print(f'internal  step(obs) B={B}: {dt * 1e3:.3f} ms -> {B / dt:.0f} samples/s')


if __name__ == '__main__':
    main()
