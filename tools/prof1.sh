#!/bin/bash
# kernel-trace profile of the bench (run on the GPU box via gpurun): writes gpurun_out/prof_<tag>/ + a text summary
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PA_BENCH_CHILD=1      # bench.py: no nested rocprofv3 child passes, no median pass (fixed step counts)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_prof_$TAG.log 2>&1
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-300
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" > gpurun_out/kernel_stats_$TAG.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
import os
print('# %srocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline   (13 steps incl. warm-up)' % ('PA_SINGLE_STREAM=1 ' if os.environ.get('PA_SINGLE_STREAM') else ''))
print('# total kernel time %.3f ms over 13 steps = %.3f ms/step' % (tot/1e6, tot/1e6/13))
print('%7s %7s %10s %10s  %s' % ('pct', 'calls', 'avg_us', 'total_ms', 'kernel'))
for r in rows[:60]:
    print('%6.2f%% %7d %10.1f %10.2f  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Name'][:120]))
PY
head -45 gpurun_out/kernel_stats_$TAG.txt
