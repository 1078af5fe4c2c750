"""The staged crop alone (pa_crop on the benchmark's batch): per-launch wall time; run under rocprofv3 --kernel-trace --stats for
the per-kernel split."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose_adv_aug_amd.data import Augmenter, DeviceBatch
from pose_adv_aug_amd.pylib import HumanAug
batch = DeviceBatch.synthetic(24, seed=0)
aug = Augmenter(seed=1)
aug.regular(batch)
p = batch.params
print('pre-downscale samples: %d of 24, rotated: %d' % (int((p[:, 2] * 200 / 256 >= 2).sum()), int((p[:, 3] != 0).sum())))
for _ in range(5):
    HumanAug.crop_batch(batch.frames, batch.params)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    HumanAug.crop_batch(batch.frames, batch.params)
torch.cuda.synchronize()
print('pa_crop: %.1f us per batch of 24' % ((time.perf_counter() - t0) / 50 * 1e6))
