#!/bin/bash
# wgrad tile-shape sweep (experiments): prints ms/step and the wgrad class times
mkdir -p gpurun_out
for cfg in "0 0" "1 0" "2 0" "3 0" "0 1" "0 2" "0 3" "1 1"; do
  set -- $cfg
  echo "TILE1=$1 TILE9=$2" 
  PA_WGRAD_TILE1=$1 PA_WGRAD_TILE9=$2 python bench.py --no-cpu-baseline --steps 20 --warmup 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['roofline']['classes']
print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if 'wgrad' in k})
"
done 2>&1 | tee gpurun_out/sweep_wgrad.log
