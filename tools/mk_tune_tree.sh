#!/bin/bash
# A copy of the working tree with a TUNING build of the library (PA_TUNING=1: the A/B environment switches of DESIGN.md section 5 compiled
# in) under tune/ (git-ignored), so that sweeps do not touch the release libraries:  bash tools/mk_tune_tree.sh && gpurun -- 'cd tune && ...'
set -e
cd "$(dirname "$0")/.."
rm -rf tune && mkdir tune
tar cf - --exclude=.git --exclude=gpurun_out --exclude='pose_adv_aug_amd/build*' --exclude=profiles --exclude=tests/golden --exclude=ab_base --exclude=tune \
    --exclude='*.md' --exclude='*.json*' --exclude=__pycache__ --exclude=.pytest_cache --exclude='*.so' . | (cd tune && tar xf -)
(cd tune/pose_adv_aug_amd/csrc && PA_TUNING=1 PA_EXTRA="$PA_EXTRA" PA_ONLY="${PA_ONLY:-}" bash build.sh)
