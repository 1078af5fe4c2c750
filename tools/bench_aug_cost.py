import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose_adv_aug_amd.stack_hg import train_step
from pose_adv_aug_amd.data import Augmenter, DeviceBatch, AugmentAhead
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
from pose_adv_aug_amd.utils.optim import RMSprop
net = create_hg(2, 1, 16, 256, default_batch=24); net.reset_parameters(seed=0); net.train()
opt = RMSprop(net); aug = Augmenter(seed=1)
batches = [DeviceBatch.synthetic(24, seed=k) for k in range(2)]
fixed = [aug.regular(b) for b in batches]
def run(n, mode):
    ahead = AugmentAhead(aug)
    if mode == 'ahead': ahead.start(batches[0])
    for i in range(n):
        if mode == 'ahead':
            data = ahead.take(); ahead.start(batches[(i + 1) % 2] if i + 1 < n else None)
        elif mode == 'inline': data = None
        else: data = fixed[i % 2]
        train_step(net, opt, aug, batches[i % 2], data=data)
for mode in ('fixed', 'ahead', 'inline', 'fixed', 'ahead', 'inline'):
    run(10, mode); torch.cuda.synchronize(); t0 = time.perf_counter(); run(40, mode); torch.cuda.synchronize()
    print(mode, '%.3f ms/step' % ((time.perf_counter() - t0) / 40 * 1e3))
