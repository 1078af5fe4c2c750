#!/bin/bash
# Knobs of the TUNING tree (tune/: tools/mk_tune_tree.sh), interleaved with the previous round's release tree (ab_base/: recipe in tools/ab_r3.sh;
# round 5 used the round-4 tree, round 6 the round-5 tree -- the label of its lines was 'round-4 tree' until the end of round 6), one box.
#   gpurun -- 'bash tools/sweep_wq.sh > gpurun_out/sweep_wq.txt 2>&1'
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor "$@" 2>/dev/null | python -c "
import json, sys
l = [l for l in sys.stdin if l.startswith('{')]
print(json.loads(l[0])['ms_per_step'] if l else 'FAILED')"; }
T=$GRAFT_REPO_ROOT/tune
run() { echo -n "$*: "; (cd $T; env $@ bash -c "$(declare -f one); one"); }
base() { echo -n "ab_base tree: "; (cd $GRAFT_REPO_ROOT/ab_base; one); }
base; run X=0
for kv in "$@"; do run $kv; done
base; run X=0
