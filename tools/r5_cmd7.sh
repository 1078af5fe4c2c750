cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PA_BENCH_CHILD=1
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_r5t -o bench -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor > gpurun_out/r5t_bench.log 2>&1
DB=$(find gpurun_out/prof_r5t -name "*results.db" | head -1)
python tools/trace_dump.py $DB 8 > gpurun_out/r5t_step.tsv
python tools/trace_gaps.py $DB 8 > gpurun_out/r5t_gaps.txt
rm -rf gpurun_out/prof_r5t
unset PA_BENCH_CHILD
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor 2>/dev/null | python -c "
import json,sys
print(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'])"; done > gpurun_out/r5t_ms.txt
