#!/bin/bash
# A/B of the persistent weight-gradient workgroup counts (slab bytes vs parallelism), one box, interleaved
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for cfg in "256 256" "128 256" "256 128" "128 128" "64 128" "128 64" "512 256" "256 512" "256 256"; do
  set -- $cfg
  echo -n "WGS1=$1 WGS9=$2: "; PA_WGRAD_WGS1=$1 PA_WGRAD_WGS9=$2 run
done
