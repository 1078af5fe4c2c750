"""What the work between two training steps costs (optimizer, bf16 weight re-pack, PCKh meters): timing variants of the bench loop."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose_adv_aug_amd import stack_hg
from pose_adv_aug_amd.data import Augmenter, DeviceBatch, AugmentAhead
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
from pose_adv_aug_amd.utils.optim import RMSprop
net = create_hg(2, 1, 16, 256, default_batch=24); net.reset_parameters(seed=0); net.train()
opt = RMSprop(net); aug = Augmenter(seed=1)
batches = [DeviceBatch.synthetic(24, seed=k) for k in range(2)]
def run(n, pckh, ahead_on=True):
    ahead = AugmentAhead(aug)
    ahead.start(batches[0])
    for i in range(n):
        data = ahead.take(); ahead.start(batches[(i + 1) % 2] if i + 1 < n else None)
        stack_hg.train_step(net, opt, aug, batches[i % 2], want_pckh=pckh, data=data)
for name, kw in (('full', dict(pckh=True)), ('no meters', dict(pckh=False)), ('full', dict(pckh=True)), ('no meters', dict(pckh=False))):
    run(10, **kw); torch.cuda.synchronize(); t0 = time.perf_counter(); run(40, **kw); torch.cuda.synchronize()
    print('%-10s %.3f ms/step' % (name, (time.perf_counter() - t0) / 40 * 1e3))
