#!/bin/bash
# kernel-trace profile of the bench (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_prof1.log 2>&1
tail -2 gpurun_out/bench_prof1.log | cut -c1-400
find gpurun_out/prof1 -name "*stats*" | head
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:40]:
    print('%6.2f%% %8d calls avg %9.1f us  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:110]))
PY
