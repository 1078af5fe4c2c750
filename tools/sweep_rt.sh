#!/bin/bash
# Runtime-knob sweep (round 5): bench.py under HIP / HSA environment settings, interleaved with the default, same box.
#   gpurun -- 'bash tools/sweep_rt.sh > gpurun_out/sweep_rt.txt 2>&1'
cd $GRAFT_REPO_ROOT
one() { timeout 150 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "
import json, sys
l = [l for l in sys.stdin if l.startswith('{')]
print(json.loads(l[0])['ms_per_step'] if l else 'FAILED')"; }
# (round 5, first run: the any-order probe -- hipExtAnyOrderLaunch shortens the gap between two kernels of one stream from 3.6 us to ~0 at
#  256 workgroups but never overlaps them -- and HIP_FORCE_DEV_KERNARG=1 (= default) / =0 (+0.5 ms) / HSA_NO_SCRATCH_RECLAIM=1 (+-0);
#  ROC_SYSTEM_SCOPE_SIGNAL=0 HANGS the process: every run is under `timeout` now)
for rep in 1; do
  echo "== rep $rep"
  echo -n "default: "; one
  for kv in DEBUG_HIP_KERNARG_COPY_OPT=0 \
            ROC_USE_FGS_KERNARG=0 HSA_ENABLE_INTERRUPT=0 GPU_STREAMOPS_CP_WAIT=0 DEBUG_HIP_DYNAMIC_QUEUES=0 AMD_DIRECT_DISPATCH=0 ROC_AQL_QUEUE_SIZE=65536 \
            HSA_ALLOCATE_QUEUE_DEV_MEM=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0; do
    echo -n "$kv: "; env $kv bash -c "$(declare -f one); one"
  done
  echo -n "default: "; one
  echo -n "graph 1: "; one --graph 1
  for kv in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 DEBUG_HIP_GRAPH_BATCH_SIZE=1; do
    echo -n "graph 1 $kv: "; env $kv bash -c "$(declare -f one); one --graph 1"
  done
  echo -n "graph 1 PACKET_CAPTURE=0 + QUEUES=4: "; env DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 bash -c "$(declare -f one); one --graph 1"
done
