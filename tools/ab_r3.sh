#!/bin/bash
# same-box comparison of the round-3 tree (r3_tmp/, built from commit aee3bb1) with the current one
cd /tmp && export TMPDIR=/tmp
one() { python bench.py --no-cpu-baseline --no-parity --no-roofline "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'])"; }
for rep in 1 2 3 4; do
cd $GRAFT_REPO_ROOT/r3_tmp; echo -n "round3 tree  c2: "; one --steps 100 --warmup 20
cd $GRAFT_REPO_ROOT;        echo -n "current     c2 fin 0: "; one --no-traffic --no-floor --fin-rows 0
cd $GRAFT_REPO_ROOT;        echo -n "current     c2 fin 128: "; one --no-traffic --no-floor
cd $GRAFT_REPO_ROOT/r3_tmp; echo -n "round3 tree  c5: "; one --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10
cd $GRAFT_REPO_ROOT;        echo -n "current     c5 fin 0: "; one --no-traffic --no-floor --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10 --fin-rows 0
cd $GRAFT_REPO_ROOT;        echo -n "current     c5 fin 128: "; one --no-traffic --no-floor --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10
done
cd $GRAFT_REPO_ROOT/r3_tmp; python tools/bench_conv3.py 2>&1 | grep -E "64x 64|32x 32" | sed 's/^/r3  /'
cd $GRAFT_REPO_ROOT; python tools/bench_conv3.py 2>&1 | grep -E "64x 64|32x 32" | sed 's/^/cur /'
