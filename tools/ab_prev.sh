#!/bin/bash
# same-box A/B of the working tree's library against build_prev/libposeadv_hip_prev.so (built from another commit)
run() { python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
cp pose_adv_aug_amd/libposeadv_hip.so /tmp/new.so
for i in 1 2; do
  cp /tmp/new.so pose_adv_aug_amd/libposeadv_hip.so; echo -n "new: "; run
  if [ -n "$AB_ENV" ]; then echo -n "new with $AB_ENV: "; env $AB_ENV python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; fi
  cp build_prev/libposeadv_hip_prev.so pose_adv_aug_amd/libposeadv_hip.so; echo -n "prev: "; run
done
cp /tmp/new.so pose_adv_aug_amd/libposeadv_hip.so
