#!/usr/bin/env python3
"""Stall attribution table from the counter passes of tools/stall_run.sh.

usage: stall_table.py <config>.json [kernel substring ...]
Units (MI355X_MICROARCH.md, rocprofv3 PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: dependency / pipe) + ACTIVE_INST_ANY
~= WAVE_CYCLES.  The SQ values are per sampled shader engine, so only RATIOS inside one kernel are meaningful."""
import json
import sys

d = json.load(open(sys.argv[1]))
pats = sys.argv[2:] or ['k_catbuild_bwd_mfma', 'k_catbuild_mfma', 'k_heads_bwd', 'k_heads_fwd', 'k_gemm_mfma_dw']


def pct(a, b):
    return f'{100.0 * a / b:5.1f}%' if b else '   n/a'


for name, c in sorted(d.items()):
    if not any(p in name for p in pats):
        continue
    g = lambda k: c.get(k, 0.0)  # noqa: E731
    wc = g('SQ_WAVE_CYCLES')
    waves = max(g('SQ_WAVES'), 1.0)
    print(f'== {name}')
    print(f'  waves (sampled) {waves:.0f}; wave life {4 * wc / waves:.0f} cycles; mean resident waves {wc / max(g("SQ_BUSY_CYCLES"), 1):.1f} per sampled SQ')
    print(f'  wave-cycle split:  issuing {pct(g("SQ_ACTIVE_INST_ANY"), wc)}   parked at s_waitcnt/barrier {pct(g("SQ_WAIT_ANY"), wc)}   '
          f'issue-stalled (dependency / busy pipe) {pct(g("SQ_WAIT_INST_ANY"), wc)}  of which LDS-issue {pct(g("SQ_WAIT_INST_LDS"), wc)}')
    print(f'  issuing, by unit:  VALU+MFMA {pct(g("SQ_ACTIVE_INST_VALU"), wc)}  scalar {pct(g("SQ_ACTIVE_INST_SCA"), wc)}  LDS {pct(g("SQ_ACTIVE_INST_LDS"), wc)}  '
          f'VMEM {pct(g("SQ_ACTIVE_INST_VMEM") + g("SQ_ACTIVE_INST_FLAT"), wc)}  misc {pct(g("SQ_ACTIVE_INST_MISC"), wc)}')
    print(f'  per wave: VALU {g("SQ_INSTS_VALU") / waves:.0f}  MFMA {g("SQ_INSTS_MFMA") / waves:.0f}  SALU {g("SQ_INSTS_SALU") / waves:.0f}  SMEM {g("SQ_INSTS_SMEM") / waves:.0f}  '
          f'LDS {g("SQ_INSTS_LDS") / waves:.0f}  VMEM rd {g("SQ_INSTS_VMEM_RD") / waves:.0f} wr {g("SQ_INSTS_VMEM_WR") / waves:.0f}  branch {g("SQ_INSTS_BRANCH") / waves:.0f}')
    ia = g('SQ_LDS_IDX_ACTIVE')
    print(f'  LDS: array-active cycles per LDS instruction {ia / max(g("SQ_INSTS_LDS"), 1):.2f}; bank-conflict share of array cycles {pct(g("SQ_LDS_BANK_CONFLICT"), ia)}; '
          f'array busy vs kernel {pct(ia, 4 * g("SQ_BUSY_CYCLES"))} (uncalibrated scale)')
    print(f'  in flight per resident wave (time-average): LDS ops {g("SQ_INST_LEVEL_LDS") / max(wc, 1):.2f}  VMEM ops {g("SQ_INST_LEVEL_VMEM") / max(wc, 1):.2f}  SMEM {g("SQ_INST_LEVEL_SMEM") / max(wc, 1):.2f}')
    hit, miss = g('TCC_HIT_sum'), g('TCC_MISS_sum')
    rq = g('TCP_TCC_READ_REQ_sum')
    print(f'  L2: hit rate {pct(hit, hit + miss)}; L1->L2 read latency {g("TCP_TCC_READ_REQ_LATENCY_sum") / max(rq, 1):.0f} cycles per request; '
          f'TA busy {pct(g("TA_TA_BUSY_sum"), 256 * g("GRBM_GUI_ACTIVE"))} of CU-cycles')
    print(f'  MFMA pipe: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x SQ_BUSY_CYCLES x CUs sampled): raw {g("SQ_VALU_MFMA_BUSY_CYCLES"):.0f} vs busy {g("SQ_BUSY_CYCLES"):.0f}')
