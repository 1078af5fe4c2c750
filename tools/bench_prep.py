"""pa_net_prepare_weights (fp32 master weights -> bf16 compute copies of the 2-stack net) alone: us per call."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose_adv_aug_amd._lib import lib, check
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
net = create_hg(2, 1, 16, 256, default_batch=24); net.reset_parameters(seed=0)
h = net._net(24)
for _ in range(5): check(lib().pa_net_prepare_weights(h))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): check(lib().pa_net_prepare_weights(h))
torch.cuda.synchronize()
print('pa_net_prepare_weights: %.1f us' % ((time.perf_counter() - t0) / 200 * 1e6))
