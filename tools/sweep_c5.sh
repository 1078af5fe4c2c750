#!/bin/bash
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10 2>/dev/null | python -c "
import json, sys
l = [l for l in sys.stdin if l.startswith('{')]
print(json.loads(l[0])['ms_per_step'] if l else 'FAILED')"; }
T=$GRAFT_REPO_ROOT/tune
run() { echo -n "$*: "; (cd $T; env $@ bash -c "$(declare -f one); one"); }
run X=0
for kv in "$@"; do run $kv; done
run X=0
