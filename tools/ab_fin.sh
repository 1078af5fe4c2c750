#!/bin/bash
# A/B of the BatchNorm finalize in the consumer's prologue (bench.py --fin-rows 0 / 128) on BASELINE configs[4] and configs[1], interleaved on one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { python bench.py --no-traffic --no-cpu-baseline --no-parity --no-floor --no-roofline "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['ms_per_step_median'])"; }
for v in 0 128 0 128; do echo -n "configs[4] 8-stack 384 bs16 fp16 --fin-rows $v: "; one --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10 --fin-rows $v; done
for v in 0 128 0 128; do echo -n "configs[1] 2-stack 256 bs24 bf16 --fin-rows $v: "; one --fin-rows $v; done
