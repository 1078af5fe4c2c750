"""Oracle (test infrastructure, see oracle/__init__.py): the per-batch work of
the reference's training loops, restated on PyTorch-CPU fp32.

  pose_train_step   stack-hg.py:153-180  (forward, sum-over-stacks MSE, backward,
                                          RMSprop, heat-map PCK)
  rmsprop_update    torch.optim.RMSprop as configured at stack-hg.py:51-52
                    (alpha .99, eps 1e-8 outside the sqrt, no momentum / decay)
  agent_logits      joint-train-pose-s-r-agent.py:250 (half-hourglass + agent)
  agent_kl_loss     joint-train-pose-s-r-agent.py:399-407
"""
import torch
import torch.nn.functional as F

from . import pylib


def rmsprop_update(p, g, v, lr, alpha=0.99, eps=1e-8):
    """In place: v <- alpha v + (1-alpha) g^2 ; p <- p - lr g / (sqrt(v) + eps)."""
    v.mul_(alpha).addcmul_(g, g, value=1 - alpha)
    p.addcdiv_(g, v.sqrt().add_(eps), value=-lr)


def make_optimizer(net, lr=2.5e-4):
    return torch.optim.RMSprop(net.parameters(), lr=lr, alpha=0.99, eps=1e-8, momentum=0, weight_decay=0)


def pose_loss_and_grads(net, img, heat):
    out = net(img)
    loss = pylib.stack_mse(out, heat)
    net.zero_grad()
    loss.backward()
    return out, loss


def pose_train_step(net, optimizer, img, heat, pck_idx=(0, 1, 2, 3, 4, 5, 10, 11, 14, 15)):
    net.train()
    out, loss = pose_loss_and_grads(net, img, heat)
    optimizer.step()
    acc = pylib.accuracy(out[-1].detach(), heat, list(pck_idx))
    return float(loss), float(acc[0]), [o.detach() for o in out]


def agent_logits(net, agent, img):
    return net(img, agent, is_half_hg=True, is_aug=True)


def agent_kl_loss(scale_logits, rot_logits, grnd_scale, grnd_rot):
    ps = F.softmax(scale_logits, dim=1)
    pr = F.softmax(rot_logits, dim=1)
    ls = F.kl_div(torch.log(ps + 1e-7), grnd_scale, reduction='mean') * grnd_scale.size(1)
    lr = F.kl_div(torch.log(pr + 1e-7), grnd_rot, reduction='mean') * grnd_rot.size(1)
    return ls + lr
