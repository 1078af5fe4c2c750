#!/usr/bin/env python3
"""Per-queue timeline of one training step from a rocprofv3 --kernel-trace of bench.py (rocpd results.db): busy time per queue,
the gap distribution of the busiest (main) queue and its largest gaps with the kernels on both sides -- where the main chain waits.

    rocprofv3 --kernel-trace --output-format rocpd -d out -o bench -- python bench.py --steps 12 --warmup 4 --no-roofline ...
    python tools/trace_gaps.py out/**/bench_results.db [step_index]"""
import collections
import sqlite3
import sys


def main(db, step=6):
    rows = list(sqlite3.connect(db).execute('select name, start, end, queue_id from kernels order by start'))
    marks = [i for i, r in enumerate(rows) if 'rmsprop' in r[0]]
    lo, hi = marks[step] + 1, marks[step + 1] + 1
    seg = rows[lo:hi]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    print('step %d: %d launches, wall %.3f ms' % (step, len(seg), (t1 - t0) / 1e6))
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r[3]].append(r)
    main_q = max(byq, key=lambda q: sum(r[2] - r[1] for r in byq[q]))
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        print('queue %s%s: %4d launches, busy %.3f ms' % (q, ' (main)' if q == main_q else '', len(rs), sum(r[2] - r[1] for r in rs) / 1e6))
    m = byq[main_q]
    gaps = [(m[i + 1][1] - m[i][2], i) for i in range(len(m) - 1)]
    tot = sum(g for g, _ in gaps)
    print('main queue: gaps total %.3f ms, median %.2f us' % (tot / 1e6, sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3))
    for thr in (1, 2, 5, 10, 20):
        sel = [g for g, _ in gaps if g > thr * 1e3]
        print('  gaps > %2d us: %3d, sum %.3f ms' % (thr, len(sel), sum(sel) / 1e6))
    print('largest gaps on the main queue:')
    for g, i in sorted(gaps, reverse=True)[:25]:
        # what ran on other queues during the gap
        others = [r[0].split('(')[0][:40] for q, rs in byq.items() if q != main_q for r in rs if r[1] < m[i + 1][1] and r[2] > m[i][2]]
        print('  %7.1f us after %-48s before %-48s | other queues: %s' % (g / 1e3, m[i][0].split('(')[0][:48], m[i + 1][0].split('(')[0][:48], ', '.join(others[:4])))
    # the step boundary: everything from 60 launches before this step's optimizer kernel to 25 after it, all queues
    print('step boundary (t in us relative to the rmsprop kernel; q = queue):')
    r0 = rows[marks[step + 1]][1]
    for r in rows[marks[step + 1] - 60: marks[step + 1] + 26]:
        print('  %9.1f .. %9.1f  q%s  %s' % ((r[1] - r0) / 1e3, (r[2] - r0) / 1e3, r[3], r[0].split('(')[0][:70]))
    # time with 1 / 2 / 3 queues busy
    ev = []
    for r in seg:
        ev.append((r[1], 1)); ev.append((r[2], -1))
    ev.sort()
    lvl, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[lvl] += t - last; last = t; lvl += d
    print('concurrency: ' + ', '.join('%d queues %.3f ms' % (k, v / 1e6) for k, v in sorted(hist.items())))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
