#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 rocpd database (the `--kernel-trace --stats` view).

usage: rocpd_summary.py results.db out.csv [steps]
`rocprofv3 --stats` on this pool writes a SQLite file; its `top_kernels` view is the stats table.
"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    c = sqlite3.connect(db)
    rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    # median / min / max from the dispatches themselves: the mean of a short run is pulled up by the first (cold) launch of every
    # kernel -- first touch of a multi-GB workspace costs milliseconds at the large configs
    durs = {}
    for name, st, en in c.execute('select name, start, end from kernels'):
        durs.setdefault(name, []).append((en - st) / 1e3)
    with open(out, 'w') as fh:
        fh.write('kernel,calls,total_us,avg_us,percent' + (',us_per_step' if steps else '') + ',median_us,min_us,max_us\n')
        for name, calls, tot, avg, pct in rows:
            d = sorted(durs.get(name, [0.0]))
            name = name.split('(')[0].replace('void ', '')
            extra = f',{tot / steps:.2f}' if steps else ''
            fh.write(f'"{name}",{calls},{tot:.1f},{avg:.3f},{pct:.2f}{extra},{d[len(d) // 2]:.3f},{d[0]:.3f},{d[-1]:.3f}\n')
        total = sum(r[2] for r in rows)
        fh.write(f'"TOTAL",{sum(r[1] for r in rows)},{total:.1f},,100.0' + (f',{total / steps:.2f}' if steps else '') + '\n')


if __name__ == '__main__':
    main()
