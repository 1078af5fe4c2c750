#!/bin/bash
# One step of a rocprofv3 --kernel-trace of bench.py as a table + per-queue summary (run on the GPU box):
#   gpurun -- 'bash tools/trace_step.sh [tag] [bench.py flags]'  ->  gpurun_out/<tag>_step.tsv, <tag>_summary.txt, <tag>_gaps.txt
TAG=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PA_BENCH_CHILD=1      # bench.py: no nested rocprofv3 child passes, no median pass
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_$TAG -o bench -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor "$@" > gpurun_out/${TAG}_bench.log 2>&1
DB=$(find gpurun_out/prof_$TAG -name "*results.db" | head -1)
python tools/trace_dump.py $DB 8 > gpurun_out/${TAG}_step.tsv
python tools/trace_summary.py gpurun_out/${TAG}_step.tsv > gpurun_out/${TAG}_summary.txt
python tools/trace_gaps.py $DB 8 > gpurun_out/${TAG}_gaps.txt
rm -rf gpurun_out/prof_$TAG
