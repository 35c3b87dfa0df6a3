"""Probe (GPU box): one mini-batch issued as K SEQUENTIAL sub-batches on ONE stream (same workspace slot, loss_scale = share;
samples are independent -- /root/reference/molgym/ppo.py:36-42 -- so the gradient accumulates exactly, tests/test_gpu_large.py) against
the one-pass step.  The question (round-5 verdict, item 3b): the rows of one level are 401 MB at cfg3 and ~2.5 GB at cfg5, larger than
the 256 MiB Infinity Cache, so every producer -> consumer hand-off goes through HBM; K chunks sized so that rows + adjoint of a level
stay cache-resident would turn re-reads into cache hits.  Time only.
usage: python tools/seq_chunk_probe.py <config> K1 K2 ...   (K = 1 is the baseline; it is run first and last)"""
import sys
import time
import torch
sys.path.insert(0, '.')
from molgym_amd.agents.covariant import CovariantAC
from molgym_amd.spaces import ActionSpace, ObservationSpace
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
Ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
cfg = CONFIGS[name]
B = cfg['batch']
dev = torch.device('cuda:0')
torch.manual_seed(0)
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'], beta=cfg['beta'],
                 device=dev, **MODEL_DEFAULTS)
d = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=0)
ac.theta.grad = torch.zeros_like(ac.theta)
n = 30 if B <= 1024 and cfg['canvas_size'] <= 20 else 10
for K in Ks + [1]:
    cuts = [round(i * B / K) for i in range(K + 1)]
    parts = [ac.prepare_batch(d['obs'][a:b], d['act'][a:b], d['logp'][a:b], d['adv'][a:b], d['ret'][a:b]) for a, b in zip(cuts, cuts[1:])]
    shares = [(b - a) / B for a, b in zip(cuts, cuts[1:])]

    def step(i):
        if i % 10 == 0:
            ac.theta.grad.zero_()
            ac.invalidate_weights()
        for k in range(K):
            ac.ppo_minibatch(parts[k], 0.2, 0.5, 0.01, loss_scale=shares[k], slot=0, epoch_cache=True)
        if (i + 1) % 10 == 0:
            ac.fold_gradients()
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{name} B={B} as {K} sequential sub-batch(es) of ~{B // K}: {dt * 1e3:.4f} ms per mini-batch -> {B / dt:.0f} samples/s', flush=True)
    del parts
