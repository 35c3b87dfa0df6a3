"""Live kernel timing and roofline accounting for bench.py.

Timing: HIP events recorded by the library on the launch stream around its dominant kernels
(mg_profile_enable / mg_profile_report, include/molgym_hip.h).  Accounting: per kernel, the ALGORITHMIC flops of one
launch in the dense-CG convention of SURVEY.md 8(d) / tools/flops.py (what `roofline.achieved` is computed from) and,
beside it, the flops the kernel actually EXECUTES (it walks the 1392-term sparse Clebsch-Gordan table instead of the
dense 25 x 25 x 25 projection), plus the algorithmic bytes it has to move.  HBM traffic per launch comes from the PMC
summary of the SAME config (profiles/pmc_<config>.json, tools/pmc_summary.py) or is null -- never another config's."""
import ctypes as C
import json
import os

import torch

from . import _lib

PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector peak == f32-input MFMA dense peak
PEAK_HBM_GBPS = 8000.0   # HBM3E, ~8 TB/s
CLOCK_HZ = 2.4e9
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CH, NLM, CG_NNZ = 10, 25, 1392
M = [1, 3, 5, 7, 9]


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


def _dense_level_flops(n):
    """forward flops of CG aggregate + CG power of ONE atom with n neighbours at a level >= 1 (dense projection)"""
    f = 0
    for l1 in range(5):
        for l2 in range(5):
            lo, hi = abs(l1 - l2), min(l1 + l2, 4)
            proj = CH * M[l1] * M[l2] * sum(M[l] for l in range(lo, hi + 1)) * 4
            f += n * CH * M[l1] * M[l2] * 8 + proj      # aggregate: kron over n neighbours + projection
            f += CH * M[l1] * M[l2] * 6 + proj          # power
    return f


def _executed_level_flops(n):
    """what the kernels execute for the same thing: 25 x 25 complex moments per channel and neighbour (8 flops each),
    then the sparse table -- per term a real x complex multiply-add (4) for the aggregate and a complex product
    plus the same (6 + 4) for the power"""
    return n * CH * NLM * NLM * 8 + CH * CG_NNZ * (4 + 10)


def catbuild_flops(natoms, adjoint):
    """(dense-convention, executed) flops of one launch over all real atoms; the adjoint costs twice the forward"""
    mul = 2 if adjoint else 1
    return (mul * sum(_dense_level_flops(int(n)) * int(n) for n in natoms),
            mul * sum(_executed_level_flops(int(n)) * int(n) for n in natoms))


def catbuild_bwd_flops(natoms):
    return catbuild_flops(natoms, True)[0]


def mfma_issued_flops(natoms, adjoint):
    """flops the matrix cores are ISSUED by one launch (16 x 16 x 4 tiles incl. zero padding): forward 32 MFMAs per
    tile of 8 neighbours and (atom, channel); adjoint 54 per tile + 26 for the power block"""
    per_tile = 54 if adjoint else 32
    n_mfma = sum(CH * ((per_tile * -(-int(n) // 8)) + (26 if adjoint else 0)) * int(n) for n in natoms)
    return n_mfma * 2 * 16 * 16 * 4


def _algorithmic_bytes(name, natoms, num_zs):
    """bytes one launch must move at least: its inputs read once + its outputs written once (f32)"""
    ta = sum(int(n) for n in natoms)
    te = sum(int(n) * int(n) for n in natoms)
    reps = ta * NLM * 2 * CH * 4                      # one level of atom representations
    edges = te * (5 * 2 * CH + 2 * NLM) * 4            # edge nets of the level + Y_lm
    # concatenated CG channels [ag | in | sq] of one level: 62 KB per atom
    cat = ta * sum(M[l] * 2 * CH * (2 * b + 1) for l, b in enumerate((5, 12, 16, 17, 15))) * 4
    if name == 'k_catbuild_mfma':        # reads the level's representations + edge nets + Y, writes the concatenated rows
        return reps + edges + cat
    if name == 'k_catbuild_bwd_mfma':    # reads their adjoint + the same inputs, writes edge / representation adjoints
        return cat + reps + edges + reps + te * 5 * 2 * CH * 4
    return None


def sources_sha16():
    """sha256 (first 16 hex digits) over the library's sources: what a PMC summary was measured on, and what runs now"""
    import hashlib
    d = os.path.join(ROOT, 'molgym_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.inc', '.h')) and f != 'cg_tables.inc':
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def _pmc_file(config):
    path = os.path.join(ROOT, 'profiles', f'pmc_{config}.json')
    return (path, json.load(open(path))) if os.path.exists(path) else (path, None)


def pmc_source(config):
    """where the line's HBM byte counts come from: they are REPLAYED from the committed rocprofv3 --pmc summary of the same config
    (tools/artifacts.sh -> tools/pmc_summary.py; counters need their own profiler passes), not measured in the bench run itself.
    `stale` says whether the kernels have changed since that summary was taken."""
    import hashlib
    path, rec = _pmc_file(config)
    if rec is None:
        return None
    meta = rec.get('_meta', {})
    now = sources_sha16()
    return {'file': os.path.relpath(path, ROOT), 'file_sha16': hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16],
            'measured_in_this_run': False, 'collected': meta.get('collected'), 'sources_sha16_at_collection': meta.get('sources_sha16'),
            'sources_sha16_now': now, 'stale': meta.get('sources_sha16') != now,
            'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (--kernel-trace only), x2 gfx950 FETCH correction'}


def _pmc_traffic(config, name):
    rec = _pmc_file(config)[1]
    if rec is None:
        return None
    if name in rec:
        return rec[name].get('hbm_bytes_per_launch')
    # the span names are the kernels' names without template arguments (k_catbuild_bwd_mfma<false> / <true>: one of them runs at a config)
    hits = [v for k, v in rec.items() if k.startswith(name + '<') and isinstance(v, dict) and v.get('hbm_bytes_per_launch') is not None]
    return hits[0]['hbm_bytes_per_launch'] if len(hits) == 1 else None


def step_hbm(config, ms_per_step):
    """whole-step HBM traffic from the PMC summary of the same config (`_step` record of tools/pmc_summary.py) against
    the live step time: {'hbm_bytes_per_step', 'hbm_gbps', 'hbm_frac'} (None without that file)"""
    rec = (_pmc_file(config)[1] or {}).get('_step')
    if not rec:
        return {'hbm_bytes_per_step': None, 'hbm_gbps': None, 'hbm_frac': None, 'hbm_bytes_source': None}
    gbps = rec['hbm_bytes_per_step'] / (ms_per_step * 1e-3) / 1e9
    return {'hbm_bytes_per_step': rec['hbm_bytes_per_step'], 'hbm_gbps': gbps, 'hbm_frac': gbps / PEAK_HBM_GBPS,
            'hbm_bytes_source': 'replayed: see roofline.traffic_source'}


def family_rooflines(per_step_ms, natoms, num_zs):
    """`roofline.families`: per kernel family {flops per step (algorithmic, ragged), us per step (live HIP events around the
    family's launches, summed), achieved TFLOP/s, fraction of the f32 peak}.  The families are the library's timing spans:
    k_gemm_rows (forward products and input adjoints issued as GEMM launches), k_gemm_dw (every weight gradient), k_heads_fwd /
    k_heads_bwd, the two CG kernels and, on the small-batch path, k_level0 / k_edge_level (the fused per-atom kernels; see
    tools/flops.py::family_flops for what is counted where).  The spans include the HIP events that bracket every launch of
    the family (a few us per launch on families of many short launches): fractions are lower bounds."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from tools.flops import family_flops
    fl = family_flops(natoms, num_zs, fused_small='k_level0' in per_step_ms)
    fl['k_catbuild_mfma'] = catbuild_flops(natoms, False)[1] * 2      # executed (sparse-table) count, two levels
    fl['k_catbuild_bwd_mfma'] = catbuild_flops(natoms, True)[1] * 2
    out = {}
    for name, ms in per_step_ms.items():
        rec = {'us_per_step': ms * 1e3}
        if name in fl and ms > 0:
            tf = fl[name] / (ms * 1e-3) / 1e12
            rec.update(flops_per_step=fl[name], achieved_tflops=tf, frac_f32_peak=tf / PEAK_F32_TFLOPS)
        out[name] = rec
    return out


def dominant_kernel_roofline(ac, batch, natoms, cfg, config_name='cfg2'):
    spans = kernel_spans(ac, batch)
    per_step = {k: v[0] * v[1] for k, v in spans.items()}
    cg = [k for k in spans if k in ('k_catbuild_mfma', 'k_catbuild_bwd_mfma')]
    name = max(cg or spans, key=lambda k: per_step[k])
    ms = spans[name][0]
    dense, executed = catbuild_flops(natoms, 'bwd' in name) if name in cg else (None, None)
    sec = ms * 1e-3
    traffic = _pmc_traffic(config_name, name)
    alg_bytes = _algorithmic_bytes(name, natoms, len(cfg['zs']))
    out = {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': PEAK_F32_TFLOPS, 'kernel': name, 'kernel_avg_ms': ms,
           'launches_per_step': spans[name][1], 'traffic': traffic, 'traffic_source': pmc_source(config_name),
           'note': 'f32 kernel: neighbour contractions on v_mfma_f32_16x16x4_f32 (dense f32 MFMA peak 157.3 TFLOP/s == '
                   'the f32 vector peak on gfx950), sparse CG projection on the vector ALUs; achieved / frac use the '
                   'DENSE-CG algorithmic count of SURVEY 8(d), achieved_executed / frac_executed the useful flops of '
                   'the sparse-table algorithm the kernel implements (MFMA tile padding not counted)',
           'span_ms_per_step': per_step,
           'families': family_rooflines(per_step, natoms, len(cfg['zs']))}
    if dense is not None:
        out.update(achieved=dense / sec / 1e12, frac=dense / sec / 1e12 / PEAK_F32_TFLOPS,
                   achieved_dense=dense / sec / 1e12, achieved_executed=executed / sec / 1e12,
                   frac_executed=executed / sec / 1e12 / PEAK_F32_TFLOPS, algorithmic_flops_per_launch=dense,
                   executed_flops_per_launch=executed,
                   mfma_issued_tflops=mfma_issued_flops(natoms, 'bwd' in name) / sec / 1e12,
                   # matrix-core utilisation: issued v_mfma_f32_16x16x4_f32 x 32 cycles per SIMD over the 1024 SIMDs
                   mfma_pipe_util=mfma_issued_flops(natoms, 'bwd' in name) / 2048 * 32 / (sec * CLOCK_HZ * 1024))
    if alg_bytes is not None:
        out['algorithmic_bytes_per_launch'] = alg_bytes
        out['algorithmic_gbps'] = alg_bytes / sec / 1e9
    if traffic is not None:  # second fraction of SURVEY 8(d): measured HBM GB/s against the 8 TB/s roof
        out['hbm_gbps'] = traffic / sec / 1e9
        out['hbm_frac'] = traffic / sec / 1e9 / PEAK_HBM_GBPS
        if alg_bytes:
            out['traffic_over_algorithmic'] = traffic / alg_bytes
    else:
        out['hbm_gbps'] = out['hbm_frac'] = None
    return out


def internal_roofline(ac, batch):
    """`roofline` object of the SchNet internal-coordinate agent (BASELINE configs[0]).  Its step is a chain of ~35 short launches
    (0.355 ms at 140 samples): the fused SchNet interactions (`k_schnet`), the fused filter networks (`k_filter`), the head MLPs
    as grouped row / column GEMM launches (`k_gemm_rows`) and the bucketed weight gradients (`k_gemm_dw`).  Every family gets its
    algorithmic flops -- 2 rows in out per Linear layer and direction it computes -- over its live HIP-event time per step
    against the f32 MFMA peak (`families`); the headline entries are those of the family with the longest span.  The fractions are
    small because these are latency-bound launches of a few hundred workgroups, which is what the object is there to show."""
    cfg = batch.cfg
    spans = kernel_spans(ac, batch)
    per_step = {k: v[0] * v[1] for k, v in spans.items()}
    if not per_step:
        return None
    B, TA, MA, ME, W, Z = cfg.B, cfg.TA, cfg.MA, cfg.ME, cfg.W, cfg.Z
    AF, LB, F, G = W // 2, W // 4, 128, 25
    NL = AF + LB
    filt = 3 * [(ME, G, F), (ME, F, F)]                                                      # filter-generating networks
    atom = 3 * [(MA, AF, F), (MA, F, AF), (MA, AF, AF)]                                      # in2f, f2out, dense
    heads = [(B, Z, W), (B, W, LB)] * 2                                                      # phi_beta, both uses
    heads += [(TA, NL, W), (TA, W, 1), (B, NL, W), (B, W, Z), (B, NL + Z, W), (B, W, 3)]     # focus, element, continuous
    heads += [(2 * B, NL, W), (2 * B, W, 1), (B, NL, W), (B, W, W), (B, W, 1)]               # kappa, critic
    mm = lambda ls: sum(2 * r * i * o for r, i, o in ls)  # noqa: E731
    nbytes = lambda ls: sum(4 * (r * i + r * o + i * o) for r, i, o in ls)  # noqa: E731
    no_dx = {(ME, G, F), (B, Z, W)}  # inputs without a gradient: the Gaussian expansion of the distances, the bag vectors
    fwd_dx = lambda ls: mm(ls) + mm([l for l in ls if l not in no_dx])  # noqa: E731
    cfconv = 3 * (2 * ME * F + 4 * ME * F)  # agg = sum y * Wf; its two adjoints
    fam_flops = {'k_gemm_dw': mm(filt + atom + heads)}
    fam_bytes = {'k_gemm_dw': nbytes(filt + atom + heads)}
    rows = list(heads)
    if 'k_filter' in per_step:
        fam_flops['k_filter'], fam_bytes['k_filter'] = fwd_dx(filt), 2 * nbytes(filt)
    else:
        rows += filt
    if 'k_schnet' in per_step:
        fam_flops['k_schnet'], fam_bytes['k_schnet'] = fwd_dx(atom) + cfconv, 2 * nbytes(atom) + 3 * 3 * 4 * ME * F
    else:
        rows += atom
    fam_flops['k_gemm_rows'], fam_bytes['k_gemm_rows'] = fwd_dx(rows), 2 * nbytes(rows)
    families = {}
    for k, ms in per_step.items():
        if k in fam_flops and ms > 0:
            tf = fam_flops[k] / (ms * 1e-3) / 1e12
            families[k] = {'us_per_step': ms * 1e3, 'flops_per_step': fam_flops[k], 'achieved_tflops': tf,
                           'frac_f32_peak': tf / PEAK_F32_TFLOPS}
    name = max(families, key=lambda k: families[k]['us_per_step']) if families else max(per_step, key=lambda k: per_step[k])
    sec = per_step[name] * 1e-3
    out = {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': PEAK_F32_TFLOPS, 'kernel': name, 'kernel_ms_per_step': per_step[name],
           'launches_per_step': spans[name][1], 'traffic': None, 'span_ms_per_step': per_step, 'families': families,
           'note': 'SchNetAC: every dense product on v_mfma_f32_16x16x4_f32 (f32 MFMA peak == the f32 vector peak on gfx950); the step '
                   'is a chain of ~35 short launches; achieved / frac are those of the family with the longest span: its algorithmic '
                   'flops (2 rows in out per layer and direction it computes) / its live HIP-event time per step'}
    if name in fam_flops:
        flops = fam_flops[name]
        out.update(achieved=flops / sec / 1e12, frac=flops / sec / 1e12 / PEAK_F32_TFLOPS, algorithmic_flops_per_step=flops,
                   algorithmic_bytes_per_step=fam_bytes[name], algorithmic_gbps=fam_bytes[name] / sec / 1e9)
    else:
        out.update(achieved=None, frac=None)
    return out
