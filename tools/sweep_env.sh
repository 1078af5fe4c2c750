#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> bench_quick with VAR=value for each value
VAR=$1; shift
for v in "$@"; do
  echo "$VAR=$v"; env $VAR=$v tools/bench_quick.sh --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids
done
