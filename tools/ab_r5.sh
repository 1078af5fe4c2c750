#!/bin/bash
# Same-box, interleaved comparison of the PREVIOUS round's tree (ab_base/, a built copy of commit ab_base/BASE_COMMIT -- round 5: the round-4 tree, round 6: the round-5 tree; see tools/ab_r3.sh for the
# recipe) with the current one.   gpurun -- 'bash tools/ab_r5.sh [pairs] > gpurun_out/ab_r5.txt 2>&1'
PAIRS=${1:-3}
cd /tmp && export TMPDIR=/tmp
one() { python bench.py --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor "$@" 2>/dev/null | python -c "
import json, sys
l = [l for l in sys.stdin if l.startswith('{')]
print(json.loads(l[0])['ms_per_step'] if l else 'FAILED')"; }
for rep in $(seq $PAIRS); do
cd $GRAFT_REPO_ROOT/ab_base; echo -n "base     c2: "; one
cd $GRAFT_REPO_ROOT;         echo -n "current  c2: "; one
done
cd $GRAFT_REPO_ROOT/ab_base; echo -n "base     c5: "; one --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10
cd $GRAFT_REPO_ROOT;         echo -n "current  c5: "; one --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10
cd $GRAFT_REPO_ROOT/ab_base; python tools/bench_cold.py 2>&1 | grep "mode 2" | sed 's/^/base /'
cd $GRAFT_REPO_ROOT;         python tools/bench_cold.py 2>&1 | grep "mode 2" | sed 's/^/cur  /'
