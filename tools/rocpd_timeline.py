"""Timeline of the LAST full step in a rocprofv3 --kernel-trace rocpd database: one line per kernel with its start
offset, duration, gap to the previous kernel's end on any queue, and the queue.  Usage: rocpd_timeline.py <db> [marker]
The step is taken between two consecutive launches of `marker` (default k_prep_counts, once per forward): the last two, or --
third argument `skip` -- the pair `skip` steps before the end (bench.py ends with 20 eagerly issued steps whose kernels are
bracketed by HIP events for the live roofline figures: their gaps are the events', not the step's)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else 'k_prep_counts'
    rows = list(db.execute('select name, start, end, queue_id, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z), '
                           'lds_size, vgpr_count from kernels order by start'))
    marks = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if len(marks) < 2 + skip:
        raise SystemExit('marker not found often enough')
    i0, i1 = marks[-2 - skip], marks[-1 - skip]
    t0 = rows[i0][1]
    last_end = t0
    busy = 0
    print(f'{"start_us":>9} {"dur_us":>8} {"gap_us":>7} q {"wgs":>6} {"lds":>6} vgpr name')
    for name, st, en, q, wgs, lds, vg in rows[i0:i1]:
        print(f'{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {(st - last_end) / 1e3:7.1f} {q} {wgs:6d} {lds:6d} {vg:4d} {name[:60]}')
        if en > last_end:
            busy += en - max(st, last_end)
            last_end = en
    print(f'step span {(rows[i1][1] - t0) / 1e3:.1f} us, union-busy {busy / 1e3:.1f} us')


if __name__ == '__main__':
    main()
