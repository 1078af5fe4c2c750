"""Timing of the joint pose/agent loop (BASELINE configs[3] shape, 1 GPU): regular step, agent-augmented pose step,
agent update (train_agent_sr), bs = 24."""
import sys, time; sys.path.insert(0, '.')
import torch
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg, create_asn
from pose_adv_aug_amd.utils.optim import RMSprop
from pose_adv_aug_amd.data import Augmenter, DeviceBatch
from pose_adv_aug_amd import joint_train_pose_s_r_agent as J
B = 24
hg = create_hg(2, 1, 16, 256, default_batch=B); hg.reset_parameters(seed=1)
agent = create_asn(256, 256, 7, 7, is_aug=True, default_batch=B); agent.reset_parameters(seed=2)
opt_hg, opt_sr = RMSprop(hg, lr=2.5e-4), RMSprop(agent, lr=5e-5)
import os
if os.environ.get('BENCH_JOINT_FIN') is not None:      # A/B: BatchNorm finalize in the consumer's prologue off (0) / row limit
    from pose_adv_aug_amd import _lib
    for net in (hg, agent):
        _lib.check(_lib.lib().pa_net_set_fin_prologue(net._net(B), int(os.environ['BENCH_JOINT_FIN'])))
aug = Augmenter(seed=3)
batch = DeviceBatch.synthetic(B, seed=4)
def timed(fn, n):
    for _ in range(8): fn()          # (also lets the teardown of a previous process on the device finish: back-to-back runs measured 16 -> 19-21 ms)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
state = {'i': 0}
def pose_pair():
    J.train_hg_step(state['i'], hg, opt_hg, agent, aug, batch, seed=0); state['i'] += 1
    J.train_hg_step(state['i'], hg, opt_hg, agent, aug, batch, seed=0); state['i'] += 1
ms_pair = timed(pose_pair, 20)
ms_agent = timed(lambda: J.train_agent_sr(batch, hg, agent, opt_sr, aug, epoch_sr=0, seed=0), 10)
print('joint loop, bs=%d: regular+agent pose step pair %.2f ms (%.0f img/s through train_hg), agent update %.2f ms'
      % (B, ms_pair, 2 * B / ms_pair * 1e3, ms_agent))
