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


def validate_tta(out_last, out_last_flipped):
    """stack-hg.py:227-229: flip the second forward's last heat map back, swap the left/right joints, average."""
    return (out_last + pylib.flip_heatmaps(out_last_flipped)) / 2


def validate_batch(net, img, heat, center, scale, rot, grnd_pts, normalizer, pck_idx=(0, 1, 2, 3, 4, 5, 10, 11, 14, 15)):
    """stack-hg.py:206-258 for one batch (net in eval mode): loss of the plain forward, flip test-time
    augmentation, PCKh in heat-map space and at the original resolution, final predictions (original image px)."""
    with torch.no_grad():
        out1 = net(img)                                                       # :215
        loss = pylib.stack_mse(out1, heat)                                    # :216-219
        out2 = net(torch.from_numpy(img.numpy()[:, :, :, ::-1].copy()))       # :222-226
        output = validate_tta(out1[-1], out2[-1])
    res = [output.shape[2], output.shape[3]]
    pckh = pylib.accuracy(output, heat, list(pck_idx))                        # :235
    pckh_o = pylib.accuracy_origin_res(output, center, scale, res, grnd_pts, normalizer, rot)   # :237-238
    preds = pylib.final_preds(output, center, scale, res, rot)                # :253
    return loss, pckh[0], pckh_o[0], preds, output


def lost_pckh_distribution(pckhs):
    """collect-scale-ditri.py:215-238 for the per-person PCKh of the K deterministic crops of ONE person (tensor [K]):
    lost = 1 - pckh, normalised; uniform when all K crops are perfect; negative entries are an error there (exit())."""
    lost = 1 - pckhs
    if float(lost.sum()) == 0:
        return torch.ones(lost.size(0)) / lost.size(0)
    assert not bool((lost < 0).any())
    return lost / lost.sum()


def pretrain_kl_loss(scale_logits, rot_logits, scale_distri, rot_distri):
    """pretrain-s-r-agent.py:175-190: F.kl_div(LogSoftmax(pred), target) * K per head (element-mean reduction), summed."""
    ls = F.kl_div(F.log_softmax(scale_logits, 1), scale_distri, reduction='mean') * scale_distri.size(1)
    lr = F.kl_div(F.log_softmax(rot_logits, 1), rot_distri, reduction='mean') * rot_distri.size(1)
    return ls + lr
