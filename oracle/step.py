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


def separated_s_r_pckh(hg, img_list, c_list, s_list, r_list, grnd_pts_list, heatmap_list, normalizer_list):
    """compute_separated_s_r_pckh, joint-train-pose-s-r-agent.py:452-467: full forward on the scale-only and the
    rotation-only crops, per-person PCKh of each."""
    assert len(img_list) == 2
    out = []
    for k in range(2):
        with torch.no_grad():
            o = hg(img_list[k])
        out.append(pylib.per_person_pckh(o[-1], heatmap_list[k], c_list[k], s_list[k], [64, 64], grnd_pts_list[k],
                                         normalizer_list[k], r_list[k]))
    return out


def train_agent_sr(hg, agent_sr, optimizer_sr, img_std, regular, agent_crops, scale_index_list, rotation_index_list,
                   pckh_override=None):
    """train_agent_sr, joint-train-pose-s-r-agent.py:317-422, ONE batch, with everything the reference draws at random
    passed in: `regular` = what the AGENT loader yields with separate_s_r (img_list, heatmap_list, c_list, s_list, r_list,
    grnd_pts_list, normalizer_list -- :326-328), the sampled bins (:352-362), and `agent_crops` = what load_batch_data
    returns for those bins (:364-369).  pckh_override = (regular_pckh_list, sr_pckh_list) replaces the four PCKh vectors
    (an untrained pose net scores ~0 everywhere; the override exercises both branches of the reward shaping).
    Returns a dict of every intermediate; the agent's parameters have taken one RMSprop step."""
    hg.eval()                                                                # :323-324
    agent_sr.train()
    keys = ('img', 'heatmap', 'c', 's', 'r', 'grnd_pts', 'normalizer')
    reg = [[d[k] for d in regular] for k in keys]
    regular_pckh_list = separated_s_r_pckh(hg, reg[0], reg[2], reg[3], reg[4], reg[5], reg[1], reg[6])      # :331-336
    ls, lr = hg(img_std, agent_sr, is_half_hg=True, is_aug=True)             # :340-342
    ps, pr = F.softmax(ls, dim=1), F.softmax(lr, dim=1)                      # :343-344
    ag = [[d[k] for d in agent_crops] for k in keys]
    sr_pckh_list = separated_s_r_pckh(hg, ag[0], ag[2], ag[3], ag[4], ag[5], ag[1], ag[6])                  # :372-374
    if pckh_override is not None:
        regular_pckh_list, sr_pckh_list = pckh_override
    idx_s = torch.as_tensor(scale_index_list).long().view(-1, 1)             # :376-377
    idx_r = torch.as_tensor(rotation_index_list).long().view(-1, 1)
    gs = pylib.gen_groundtruth(ps.detach(), idx_s, regular_pckh_list[0], sr_pckh_list[0])                   # :379-389
    gr = pylib.gen_groundtruth(pr.detach(), idx_r, regular_pckh_list[1], sr_pckh_list[1])
    loss_scale = F.kl_div(torch.log(ps + 1e-7), gs, reduction='mean') * gs.size(1)                          # :399-404
    loss_rot = F.kl_div(torch.log(pr + 1e-7), gr, reduction='mean') * gr.size(1)
    loss = loss_scale + loss_rot
    optimizer_sr.zero_grad()
    loss.backward()
    optimizer_sr.step()                                                      # :408-410
    return dict(pckh_regular=regular_pckh_list, pckh_agent=sr_pckh_list, logits=(ls.detach(), lr.detach()),
                probs=(ps.detach(), pr.detach()), targets=(gs, gr), loss=loss.detach())
