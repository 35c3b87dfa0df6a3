#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs).

usage: pmc_summary.py fetch_results.db write_results.db out.json [sq_results.db]
(the optional third database holds an SQ pass: SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32, SQ_BUSY_CU_CYCLES,
GRBM_GUI_ACTIVE -> raw per-launch averages per kernel plus `mfma_busy_over_gpu_active` = SQ_VALU_MFMA_BUSY_CYCLES /
GRBM_GUI_ACTIVE.  The SQ counters are not chip-wide SIMD-cycle sums on this stack (for k_catbuild_bwd_mfma they come out
~1/32 of instructions x 32 cycles), so bench.py reports the matrix-core utilisation analytically -- issued MFMAs x 32
cycles / (kernel time x clock x 1024 SIMDs) -- and these raw numbers are kept next to it as the measured cross-check:
the RATIO between kernels is meaningful, the absolute scale is not calibrated.)
Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced reads, so the read side is
doubled (an upper bound for narrow accesses; WRITE_SIZE is uncalibrated and taken as is).
"""
import json
import os
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(counter_value) from pmc_events where counter_name = ? group by name',
                     (counter, )).fetchall()
    return {n.split('(')[0].replace('void ', ''): (cnt, tot) for n, cnt, tot in rows}


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for name in sorted(set(fetch) | set(write)):
        fc, ft = fetch.get(name, (0, 0.0))
        wc, wt = write.get(name, (0, 0.0))
        rd = 2.0 * ft * 1024 / max(fc, 1)
        wr = wt * 1024 / max(wc, 1)
        out[name] = {'launches': fc, 'fetch_kib_per_launch_raw': ft / max(fc, 1), 'write_kib_per_launch_raw': wt / max(wc, 1),
                     'hbm_bytes_per_launch': rd + wr}
    if len(sys.argv) > 4:
        c = sqlite3.connect(sys.argv[4])
        rows = c.execute('select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name').fetchall()
        sq = {}
        for n, cn, cnt, tot in rows:
            sq.setdefault(n.split('(')[0].replace('void ', ''), {})[cn] = (cnt, tot)
        for name, d in sq.items():
            rec = out.setdefault(name, {})
            for cn, (cnt, tot) in d.items():
                rec[cn + '_per_launch'] = tot / max(cnt, 1)
            busy, act = d.get('SQ_VALU_MFMA_BUSY_CYCLES'), d.get('GRBM_GUI_ACTIVE')
            if busy and act and act[1] > 0:
                rec['mfma_busy_over_gpu_active'] = busy[1] / act[1]
    # whole step: every kernel's bytes x launches over the steps the profiled command ran (PMC_STEPS: timed + warm-up steps
    # + the 20 iterations of bench.py's live roofline leg)
    steps = int(os.environ.get('PMC_STEPS', '0'))
    if steps > 0:
        total = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in out.values() if 'hbm_bytes_per_launch' in v)
        out['_step'] = {'steps_in_run': steps, 'hbm_bytes_per_step': total / steps}
    # what the counters were measured on (bench.py's roofline.traffic_source compares it with the sources of the run that replays them)
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from molgym_amd.profile import sources_sha16
    out['_meta'] = {'sources_sha16': sources_sha16(), 'collected': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())}
    json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    top = sorted((kv for kv in out.items() if isinstance(kv[1], dict) and 'hbm_bytes_per_launch' in kv[1] and 'launches' in kv[1]),
                 key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]
    for k, v in top:
        print(f"{k:40s} launches={v['launches']:5d} bytes/launch={v['hbm_bytes_per_launch'] / 1e6:9.3f} MB")


if __name__ == '__main__':
    main()
