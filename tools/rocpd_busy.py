"""GPU occupancy of a window of a rocprofv3 --kernel-trace rocpd database: union-busy time over all queues, the idle gaps
(> 15 us) and the kernel that ends each of them -- is the device waiting for the host or working?
usage: rocpd_busy.py <db> [window_ms from the end, default 25]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 25e6
rows = list(db.execute('select name, start, end, queue_id from kernels order by start'))
t1 = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t1 - win]
t0 = rows[0][1]
busy, last_end, gaps = 0, t0, []
per_q = {}
for name, st, en, q in rows:
    per_q[q] = per_q.get(q, 0) + (en - st)
    if st > last_end and st - last_end > 15e3:
        gaps.append(((st - last_end) / 1e3, (st - t0) / 1e3, name.split('(')[0][:40]))
    if en > last_end:
        busy += en - max(st, last_end)
        last_end = en
span = last_end - t0
print(f'window {span / 1e6:.2f} ms, {len(rows)} kernels, union-busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), '
      f'sum of kernel durations {sum(per_q.values()) / 1e6:.2f} ms over {len(per_q)} queues')
print('idle gaps > 15 us: count', len(gaps), 'total', round(sum(g[0] for g in gaps) / 1e3, 2), 'ms; the largest:')
for g in sorted(gaps, reverse=True)[:12]:
    print(f'  {g[0]:8.1f} us idle before {g[2]} at +{g[1] / 1e3:.2f} ms')
