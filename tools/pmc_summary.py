#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs).

usage: pmc_summary.py fetch_results.db write_results.db out.json
Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced reads, so the read side is
doubled (an upper bound for narrow accesses; WRITE_SIZE is uncalibrated and taken as is).
"""
import json
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
    json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    top = sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]
    for k, v in top:
        print(f"{k:40s} launches={v['launches']:5d} bytes/launch={v['hbm_bytes_per_launch'] / 1e6:9.3f} MB")


if __name__ == '__main__':
    main()
