#!/bin/bash
# Interleaved A/B of two tuning trees (tune/ = tools/mk_tune_tree.sh, tune_p/ = the same with PA_EXTRA=-D...): compile-time experiments of round 5
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-traffic --no-floor 2>/dev/null | python -c "
import json, sys
l = [l for l in sys.stdin if l.startswith('{')]
print(json.loads(l[0])['ms_per_step'] if l else 'FAILED')"; }
for i in 1 2 3 4; do
  echo -n "base:   "; (cd $GRAFT_REPO_ROOT/tune; one)
  echo -n "variant:"; (cd $GRAFT_REPO_ROOT/tune_p; one)
done
