"""Where the fused low-resolution launch (csrc/lowres_fused.hip) spends workgroup 0's time: cycles per phase and map size."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose_adv_aug_amd._lib import lib, check, ptr
from pose_adv_aug_amd.stack_hg import train_step
from pose_adv_aug_amd.data import Augmenter, DeviceBatch
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
from pose_adv_aug_amd.utils.optim import RMSprop
B = 24
net = create_hg(2, 1, 16, 256, default_batch=B); net.reset_parameters(seed=0); net.train()
opt = RMSprop(net); aug = Augmenter(seed=1)
batch = DeviceBatch.synthetic(B, seed=0)
for _ in range(3):
    train_step(net, opt, aug, batch)
t = torch.zeros(24, dtype=torch.int64, device='cuda')
check(lib().pa_net_lowres_timing(net._net(B), ptr(t)))
N = 10
for _ in range(N):
    train_step(net, opt, aug, batch)
torch.cuda.synchronize()
check(lib().pa_net_lowres_timing(net._net(B), None))
t = t.cpu().view(3, 8).double() / (N * 2)        # per launch (two stacks per step)
names = ['consts', 'stage', 'mfma+epi', 'publish', 'wait', 'collect', 'pool/up', 'drain']
print('cycles per launch (workgroup 0), shader clock ~2.1-2.4 GHz')
print('%-6s' % 'map' + ''.join('%10s' % n for n in names) + '%10s' % 'sum')
for i, m in enumerate(('16x16', '8x8', '4x4')):
    print('%-6s' % m + ''.join('%10.0f' % v for v in t[i]) + '%10.0f' % t[i].sum())
print('total %.0f cycles = %.1f us at 2.2 GHz' % (t.sum(), t.sum() / 2200))
