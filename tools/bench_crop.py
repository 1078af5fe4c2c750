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
if len(sys.argv) > 1 and sys.argv[1] == 'cold':        # the frames are not in any cache (as inside the training step): 640 MB of other traffic between calls
    scratch = torch.empty(640 << 20, dtype=torch.uint8, device='cuda')
    tot = 0.0
    for i in range(20):
        scratch.fill_(i & 1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); HumanAug.crop_batch(batch.frames, batch.params); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    print('pa_crop cold: %.1f us per batch of 24' % (tot / 20 * 1e3))
