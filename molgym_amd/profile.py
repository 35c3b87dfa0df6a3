"""Live kernel timing for bench.py's roofline object (HIP events on the launch stream)."""
import ctypes as C

import torch

from . import _lib

PEAK_F32_TFLOPS = 157.3


def time_step_stages(ac, batch, iters=20):
    """Average device time (ms) of forward and of backward, HIP events on the current stream."""
    lib = _lib.lib()
    out = ac.forward_batch(batch)
    ws = ac._last_ws
    gout = torch.ones_like(out) / batch.cfg.B
    grad = torch.zeros_like(ac.theta)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    tf = tb = 0.0
    for _ in range(iters):
        ev[0].record()
        ac.forward_batch(batch)
        ev[1].record()
        _lib.check(lib.mg_cov_backward(C.byref(batch.cfg), p(ac.theta), p(batch.pos), p(batch.charges), p(batch.bags),
                                       p(batch.actions), p(ac.leb), p(ws), ws.numel(), p(gout), p(grad), stream))
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    return tf / iters, tb / iters


def dominant_kernel_roofline(ac, batch, natoms, cfg):
    """Placeholder until per-kernel timing lands: device time of the whole fwd+bwd launch sequence."""
    from tools.flops import step_flops
    tf, tb = time_step_stages(ac, batch)
    flops = step_flops(natoms, len(cfg['zs']))
    achieved = flops / ((tf + tb) * 1e-3) / 1e12
    return {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_F32_TFLOPS, 'traffic': None,
            'kernel': 'whole fwd+bwd launch sequence (f32 VALU path; f32 MFMA peak == f32 vector peak)',
            'fwd_ms': tf, 'bwd_ms': tb}
