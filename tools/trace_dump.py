#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace (rocpd results.db) as a table: start (us from the step's first launch), duration, queue,
grid (workgroups), kernel -- for offline analysis of the schedule (tools/trace_table.py, tools/trace_gaps.py read the same database).

    python tools/trace_dump.py out/**/bench_results.db [step_index] > step.tsv"""
import sqlite3
import sys


def main(db, step=6):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
    want = ['name', 'start', 'end', 'queue_id'] + [c for c in ('grid_x', 'grid_size_x', 'workgroup_x', 'workgroup_size_x') if c in cols]
    rows = list(con.execute('select %s from kernels order by start' % ', '.join(want)))
    marks = [i for i, r in enumerate(rows) if 'rmsprop' in r[0]]
    seg = rows[marks[step] + 1: marks[step + 1] + 1]
    t0 = seg[0][1]
    print('# columns: ' + ', '.join(want))
    for r in seg:
        extra = '\t'.join(str(x) for x in r[4:])
        print('%.2f\t%.2f\t%s\t%s\t%s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], extra, r[0].split('(')[0][:110]))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
