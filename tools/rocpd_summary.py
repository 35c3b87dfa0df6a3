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
    with open(out, 'w') as fh:
        fh.write('kernel,calls,total_us,avg_us,percent' + (',us_per_step' if steps else '') + '\n')
        for name, calls, tot, avg, pct in rows:
            name = name.split('(')[0].replace('void ', '')
            extra = f',{tot / steps:.2f}' if steps else ''
            fh.write(f'"{name}",{calls},{tot:.1f},{avg:.3f},{pct:.2f}{extra}\n')
        total = sum(r[2] for r in rows)
        fh.write(f'"TOTAL",{sum(r[1] for r in rows)},{total:.1f},,100.0' + (f',{total / steps:.2f}' if steps else '') + '\n')


if __name__ == '__main__':
    main()
