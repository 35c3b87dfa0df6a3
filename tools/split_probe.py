"""Probe (GPU box): one 140-sample mini-batch issued as K concurrent sub-batches (own streams, workspaces, graph slots, loss_scale =
share) against the one-launch step: ms per mini-batch, strict (one mini-batch at a time).  usage: python tools/split_probe.py [config]"""
import sys
import time
import torch
sys.path.insert(0, '.')
from molgym_amd.agents.covariant import CovariantAC
from molgym_amd.spaces import ActionSpace, ObservationSpace
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg = CONFIGS[name]
B = cfg['batch']
dev = torch.device('cuda:0')
torch.manual_seed(0)
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'], beta=cfg['beta'],
                 device=dev, **MODEL_DEFAULTS)
d = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=0)
ac.theta.grad = torch.zeros_like(ac.theta)
for K in (1, 2, 3, 4, 2, 1):
    cuts = [round(i * B / K) for i in range(K + 1)]
    parts = [ac.prepare_batch(d['obs'][a:b], d['act'][a:b], d['logp'][a:b], d['adv'][a:b], d['ret'][a:b]) for a, b in zip(cuts, cuts[1:])]
    shares = [(b - a) / B for a, b in zip(cuts, cuts[1:])]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(K - 1)]
    ev0 = torch.cuda.Event()
    evs = [torch.cuda.Event() for _ in range(K)]

    def step(i):
        if i % 10 == 0:
            ac.theta.grad.zero_()
            ac.invalidate_weights()
        ev0.record(streams[0])
        for k in range(K):
            if k:
                streams[k].wait_event(ev0)
            with torch.cuda.stream(streams[k]):
                ac.ppo_minibatch(parts[k], 0.2, 0.5, 0.01, loss_scale=shares[k], slot=k, epoch_cache=True)
                if k:
                    evs[k].record(streams[k])
        for k in range(1, K):
            streams[0].wait_event(evs[k])
        if (i + 1) % 10 == 0:
            ac.fold_gradients()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{name} B={B} as {K} concurrent sub-batch(es): {dt * 1e3:.4f} ms per mini-batch -> {B / dt:.0f} samples/s')
