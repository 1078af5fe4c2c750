"""Per-launch-shape table of a rocprofv3 --kernel-trace CSV: kernel (template arguments kept) x grid x workgroup size ->
launches per step, average duration, ms per step.  usage: trace_table.py <kernel_trace.csv> <steps> [min_us_per_step]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:100]


def main(path, steps, floor=0.0):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
        grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, wg)
        a = agg[(short(r['Kernel_Name']), grid, wg)]
        a[0] += 1; a[1] += d
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print('# kernel time %.3f ms per step over %d steps' % (tot / 1e3 / steps, steps))
    print('%8s %8s %9s %7s %5s  %s' % ('us/step', 'n/step', 'avg_us', 'wgs', 'thr', 'kernel'))
    for (k, grid, wg), (n, d) in rows:
        if d / steps < floor:
            continue
        print('%8.1f %8.1f %9.2f %7d %5d  %s' % (d / steps, n / steps, d / n, grid, wg, k))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
