"""How long does the host take to ENQUEUE one training step (no sync) vs the GPU to run it?"""
import sys, time; sys.path.insert(0, '.')
import torch
from pose_adv_aug_amd.stack_hg import train_step
from pose_adv_aug_amd.data import Augmenter, DeviceBatch
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
from pose_adv_aug_amd.utils.optim import RMSprop
B = 24
net = create_hg(2, 1, 16, 256, res=256, default_batch=B); net.reset_parameters(seed=0)
opt = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8); aug = Augmenter(seed=1)
batch = DeviceBatch.synthetic(B, seed=0); net.train()
for _ in range(5): train_step(net, opt, aug, batch)
torch.cuda.synchronize()
for n in (1, 5, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): train_step(net, opt, aug, batch)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('steps %2d: host enqueue %.3f ms/step, wall %.3f ms/step' % (n, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
