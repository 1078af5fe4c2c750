"""Race screen: N training steps from the same initial state must give BITWISE identical parameters with the engine's
side streams on and off, and from run to run (every reduction of the engine is order-deterministic)."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import torch
from pose_adv_aug_amd import _lib
from pose_adv_aug_amd.stack_hg import train_step
from pose_adv_aug_amd.data import Augmenter, DeviceBatch
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
from pose_adv_aug_amd.utils.optim import RMSprop
B, steps = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 8
def run(multi):
    net = create_hg(2, 1, 16, 256, res=256, default_batch=B); net.reset_parameters(seed=0)
    opt = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8); aug = Augmenter(seed=1)
    batches = [DeviceBatch.synthetic(B, seed=k) for k in range(2)]
    net.train()
    _lib.check(_lib.lib().pa_net_set_multi_stream(net._net(B), 1 if multi else 0))
    losses = []
    for i in range(steps):
        loss, _, _ = train_step(net, opt, aug, batches[i % 2])
        losses.append(float(loss))
    torch.cuda.synchronize()
    return net.flat_params.clone(), net.flat_buffers.clone(), losses
a = run(True); b = run(True); c = run(False)
print('losses', [round(x, 6) for x in a[2]])
print('multi vs multi : params equal', bool(torch.equal(a[0], b[0])), 'buffers equal', bool(torch.equal(a[1], b[1])), 'max diff', float((a[0] - b[0]).abs().max()))
print('multi vs single: params equal', bool(torch.equal(a[0], c[0])), 'buffers equal', bool(torch.equal(a[1], c[1])), 'max diff', float((a[0] - c[0]).abs().max()))
