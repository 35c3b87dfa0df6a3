"""SchNetAC (BASELINE configs[0]: model=internal, canvas_size=7, mini_batch=140) forward + loss + backward
samples/s on the GPU, the ragged batch resident in HBM, next to the float64->float32 oracle on the host cores.
usage: python tools/internal_bench.py [batch] [steps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from molgym_amd.agents.internal import SchNetAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import make_batch_internal  # noqa: E402

ZS, N = [0, 9, 16], 7


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 140
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    torch.manual_seed(0)
    ac = SchNetAC(ObservationSpace(N, ZS), ActionSpace(ZS), (0.8, 1.8), 128, device='cuda:0')
    data = make_batch_internal(B, N, ZS, seed=0)
    batch = ac.make_batch(data['obs'], data['act'])
    dev = ac.theta.device
    f64 = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64)).to(dev)
    logp, adv, ret = f64(data['logp']), f64(data['adv']), f64(data['ret'])
    ac.theta.grad = torch.zeros_like(ac.theta)

    def step():
        ac.theta.grad.zero_()
        return ac.ppo_minibatch(batch, logp, adv, ret, 0.2, 0.5, 0.01)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stats = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(stats).all()
    print(f'SchNetAC fwd+bwd B={B}: {dt * 1e3:.3f} ms/step -> {B / dt:.0f} samples/s')
    # host baseline: the oracle restatement, float32, 8 threads
    from oracle.internal_ref import SchNetACRef
    from oracle.ppo_ref import compute_loss_ref
    torch.set_num_threads(8)
    ref = SchNetACRef(ZS, N, (0.8, 1.8), 128)
    ref.load_state_dict({k: v.float().cpu() for k, v in ac.export_state_dict().items()}, strict=True)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 15 and n < 10:
        ref.zero_grad()
        loss, _ = compute_loss_ref(ref, data, 0.2, 0.5, 0.01)
        loss.backward()
        n += 1
    cdt = (time.perf_counter() - t0) / n
    print(f'CPU port (8 threads, {n} passes): {cdt * 1e3:.1f} ms/step -> {B / cdt:.0f} samples/s; ratio {cdt / dt:.0f}x')


if __name__ == '__main__':
    main()
