"""Live kernel timing for bench.py's roofline object: HIP events recorded by the library on the launch
stream around its dominant kernels (mg_profile_enable / mg_profile_report, include/molgym_hip.h)."""
import ctypes as C
import json
import os

import torch

from . import _lib

PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector peak == f32-input MFMA dense peak
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_spans(ac, batch, iters=20):
    """{span name: (avg ms per launch, launches per step)} over `iters` forward+backward steps."""
    lib = _lib.lib()
    lib.mg_profile_enable(1)
    for _ in range(iters):
        ac.theta.grad = None
        ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16)
    _lib.check(lib.mg_profile_report(buf, len(buf)))
    lib.mg_profile_enable(0)
    out = {}
    for line in buf.value.decode().splitlines():
        name, total, count = line.split()
        if int(count):
            out[name] = (float(total) / int(count), int(count) / iters)
    return out


def catbuild_bwd_flops(natoms):
    """Algorithmic flops of ONE k_catbuild_bwd launch (adjoint of CG aggregate + CG power + pass-through of one
    level >= 1 over all real atoms): 2 x the forward count of tools/flops.py for that level."""
    L, C_ = 4, 10
    M = [2 * l + 1 for l in range(L + 1)]
    tot = 0
    for n in natoms:
        n = int(n)
        f = 0
        for l1 in range(L + 1):
            for l2 in range(L + 1):
                lo, hi = abs(l1 - l2), min(l1 + l2, L)
                proj = C_ * M[l1] * M[l2] * sum(M[l] for l in range(lo, hi + 1)) * 4
                f += n * (n * C_ * M[l1] * M[l2] * 8 + proj)   # aggregate: kron over n neighbours + projection
                f += n * (C_ * M[l1] * M[l2] * 6 + proj)        # power
        tot += 2 * f
    return tot


def dominant_kernel_roofline(ac, batch, natoms, cfg):
    spans = kernel_spans(ac, batch)
    per_step = {k: v[0] * v[1] for k, v in spans.items()}
    name = 'k_catbuild_bwd'
    ms = spans[name][0]
    flops = catbuild_bwd_flops(natoms)
    achieved = flops / (ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    if os.path.exists(pmc):
        traffic = json.load(open(pmc)).get(name, {}).get('hbm_bytes_per_launch')
    return {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_F32_TFLOPS, 'traffic': traffic, 'kernel': name,
            'kernel_avg_ms': ms, 'launches_per_step': spans[name][1], 'algorithmic_flops_per_launch': flops,
            'note': 'f32 VALU kernel; the f32-input MFMA dense peak equals the f32 vector peak on gfx950',
            'span_ms_per_step': per_step}
