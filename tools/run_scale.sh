#!/bin/bash
# Multi-GPU scaling run for the day an 8-GPU node is available (BASELINE configs[2..4]; nothing here needs more than the GPUs present):
#   bash tools/run_scale.sh [tag]
#   * `python bench.py --gpus N` for N = 1, 2, 4, 8 (bench.py starts its own ranks: one process per GPU, RCCL = backend nccl, file-store
#     rendezvous) with NCCL_DEBUG=INFO; the JSON lines go to gpurun_out/<tag>_scale.jsonl, the communicator facts RCCL logs (ranks,
#     ring / tree channels, transport: xGMI P2P vs SHM) to gpurun_out/<tag>_rccl_topology.txt -- copy both into profiles/;
#   * the 2-rank engine tests through RCCL (tests/test_gpu_dist.py with POSEADV_TEST_DIST_BACKEND=nccl: one rank per GPU) when >= 2 GPUs are visible;
#   * the overlapped exchange (--overlap 1) next to the single all-reduce at the largest N.
# N above the visible GPU count is skipped, never faked.
TAG=${1:-round4}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "# visible GPUs: $NGPU" | tee gpurun_out/${TAG}_rccl_topology.txt
: > gpurun_out/${TAG}_scale.jsonl
LAST=1
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "# --gpus $N skipped: $NGPU GPU(s) visible" | tee -a gpurun_out/${TAG}_rccl_topology.txt; continue; fi
  LAST=$N
  NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH python bench.py --gpus $N --steps 50 --warmup 10 --no-cpu-baseline --no-traffic \
      > gpurun_out/${TAG}_scale_$N.out 2> gpurun_out/${TAG}_scale_$N.err
  echo "# --gpus $N rc=$?" | tee -a gpurun_out/${TAG}_rccl_topology.txt
  grep '^{' gpurun_out/${TAG}_scale_$N.out >> gpurun_out/${TAG}_scale.jsonl
  grep -hE 'NCCL INFO (comm|Channel|Ring|Trees|Connected|nranks|Using network|.*via P2P|.*via SHM|.*XGMI)' gpurun_out/${TAG}_scale_$N.out gpurun_out/${TAG}_scale_$N.err \
      | sed -E 's/^[^ ]+:[0-9]+:[0-9]+ \[[0-9]+\] //' | sort | uniq -c | sort -rn | head -40 >> gpurun_out/${TAG}_rccl_topology.txt
done
if [ "$LAST" -gt 1 ]; then
  python bench.py --gpus $LAST --steps 50 --warmup 10 --no-cpu-baseline --no-traffic --overlap 1 | grep '^{' >> gpurun_out/${TAG}_scale.jsonl
fi
if [ "$NGPU" -ge 2 ]; then
  POSEADV_TEST_DIST_BACKEND=nccl python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/${TAG}_dist_nccl.txt 2>&1
  tail -3 gpurun_out/${TAG}_dist_nccl.txt
else
  echo "# tests/test_gpu_dist.py over RCCL skipped: needs >= 2 GPUs" | tee -a gpurun_out/${TAG}_rccl_topology.txt
fi
python - $TAG <<'PY'
import json, sys
rows = [json.loads(l) for l in open('gpurun_out/%s_scale.jsonl' % sys.argv[1]) if l.startswith('{')]
for r in rows:
    print('%d GPU(s) %-28s %9.1f img/s  %.3f ms/step' % (r['n_gpus'], r['config']['parallelism'], r['value'], r['ms_per_step']))
PY
