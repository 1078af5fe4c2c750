#!/bin/bash
# Same-box, interleaved comparison of an older tree (ab_base/: round 4 used the round-3 tree, round 5 the round-4 tree -- tools/ab_r5.sh is the shorter form) with the current one (boxes of the pool differ by up to 3 % on the same binary,
# and an in-build A/B cannot see a regression that hits both of its arms).  ab_base/ is NOT in git; make it with
#   git worktree add /tmp/r3 aee3bb1 && (cd /tmp/r3 && bash pose_adv_aug_amd/csrc/build.sh)
#   mkdir ab_base && (cd /tmp/r3 && tar cf - --exclude=.git --exclude=gpurun_out --exclude='pose_adv_aug_amd/build*' --exclude=profiles --exclude=tests/golden .) | (cd ab_base && tar xf -)
# and run through gpurun:  gpurun -- 'bash tools/ab_r3.sh'
cd /tmp && export TMPDIR=/tmp
one() { python bench.py --no-cpu-baseline --no-parity --no-roofline "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'])"; }
for rep in 1 2 3 4; do
cd $GRAFT_REPO_ROOT/ab_base; echo -n "round3 tree  c2: "; one --steps 100 --warmup 20
cd $GRAFT_REPO_ROOT;        echo -n "current     c2 fin 0: "; one --no-traffic --no-floor --fin-rows 0
cd $GRAFT_REPO_ROOT;        echo -n "current     c2 fin 128: "; one --no-traffic --no-floor
cd $GRAFT_REPO_ROOT/ab_base; echo -n "round3 tree  c5: "; one --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10
cd $GRAFT_REPO_ROOT;        echo -n "current     c5 fin 0: "; one --no-traffic --no-floor --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10 --fin-rows 0
cd $GRAFT_REPO_ROOT;        echo -n "current     c5 fin 128: "; one --no-traffic --no-floor --stacks 8 --res 384 --bs 16 --dtype fp16 --steps 40 --warmup 10
done
cd $GRAFT_REPO_ROOT/ab_base; python tools/bench_conv3.py 2>&1 | grep -E "64x 64|32x 32" | sed 's/^/r3  /'
cd $GRAFT_REPO_ROOT; python tools/bench_conv3.py 2>&1 | grep -E "64x 64|32x 32" | sed 's/^/cur /'
