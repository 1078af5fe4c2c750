cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5o.txt
bash tools/ab_r5.sh 2 > gpurun_out/ab_r5b.txt 2>&1
bash tools/r5_cmd7.sh
# graph replay: which queues does the runtime use for the captured step?
export PA_BENCH_CHILD=1
for v in default queues4; do
  if [ $v = queues4 ]; then export DEBUG_HIP_FORCE_GRAPH_QUEUES=4; fi
  rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_r5g_$v -o bench -- python bench.py --graph 1 --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor > gpurun_out/r5g_${v}_bench.log 2>&1
  DB=$(find gpurun_out/prof_r5g_$v -name "*results.db" | head -1)
  python tools/trace_dump.py $DB 8 > gpurun_out/r5g_${v}_step.tsv
  rm -rf gpurun_out/prof_r5g_$v
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES PA_BENCH_CHILD
python tools/cpu_thread_sweep.py > gpurun_out/cpu_thread_sweep.json 2> gpurun_out/cpu_thread_sweep.err
