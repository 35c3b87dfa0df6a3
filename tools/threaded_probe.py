"""Probe (GPU box): the mini-batches of one epoch issued round-robin on three streams from ONE host thread (what ppo.train did)
vs from THREE host threads, one per stream / workspace slot (ctypes releases the GIL inside the library calls).
Usage: python tools/threaded_probe.py [config] [steps]"""
import sys
import threading
import time

import torch

sys.path.insert(0, '.')
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    cfg = CONFIGS[name]
    dev = torch.device('cuda')
    torch.manual_seed(0)
    ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'],
                     beta=cfg['beta'], device=dev, **MODEL_DEFAULTS)
    data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=0)
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    ac.theta.grad = torch.zeros_like(ac.theta)
    nstream = 3
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]

    def single(n):
        for i in range(n):
            k = i % nstream
            with torch.cuda.stream(streams[k]):
                ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, slot=k)

    def worker(k, n):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(streams[k]):
                for _ in range(n):
                    ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, slot=k)
        except Exception:  # noqa
            import traceback
            traceback.print_exc()

    def threaded(n):
        ts = [threading.Thread(target=worker, args=(k, n // nstream)) for k in range(nstream)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    for label, fn in (('one thread', single), ('three threads', threaded), ('one thread', single), ('three threads', threaded)):
        fn(30)
        torch.cuda.synchronize()
        ac.theta.grad.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g = float(ac.theta.grad.double().abs().sum())
        print(f'{label}: {dt / steps * 1e3:.4f} ms per mini-batch ({cfg["batch"] * steps / dt:.0f} samples/s), host issue '
              f'{t_issue / steps * 1e3:.4f} ms, |grad|_1 {g:.6e}')


if __name__ == '__main__':
    main()
