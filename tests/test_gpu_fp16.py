"""BASELINE configs[4] (8-stack 384x384, fp16 operands): the IEEE-half build of the library against the fp32 oracle.
The storage type is per process (pose_adv_aug_amd._lib.DTYPE), so the checks live in tests/fp16_check.py and run in a child."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp16_build_matches_the_oracle():
    env = dict(os.environ, POSEADV_DTYPE='fp16')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'fp16_check.py')], env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and 'ALL FP16 CHECKS PASSED' in r.stdout, tail
    assert 'FAIL' not in r.stdout, tail
