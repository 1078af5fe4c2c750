"""joint-train-pose-s-r-agent.py of the reference (stage 3: adversarial scale/rotation augmentation) on the
HIP engine.  The DataLoader that the reference re-forks for every agent batch (load_batch_data, :425-450)
becomes a re-warp of the resident frames with the sampled bins: the agent's augmentations never touch the host.

  train_hg        :195-315   even steps: regular augmentation; odd steps: half-hourglass (pose net in TRAIN
                             mode, Appendix A.9) + agent (eval) -> softmax -> categorical draw per sample ->
                             agent-law crop -> full pose step
  train_agent_sr  :317-422   pose net in eval mode, agent in train mode; per-person PCKh of regular
                             scale-only / rotation-only crops vs the agent's; reward-shaped targets
                             (utils/util.gen_groundtruth); KL loss; ONE batch per call (Appendix A.8)
"""
from collections import OrderedDict

import torch

from ._lib import lib, check, ptr, stream
from .stack_hg import PCK_IDX
from .utils import util
from .utils.util import AverageMeter


def sample_bins(logits, seed, step, slot):
    """softmax + np.random.choice(K, p) per sample (:252-271) on the device.  Returns (probs [B][K], idx int32 [B])."""
    B, K = logits.shape
    probs = torch.empty_like(logits)
    idx = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(lib().pa_sample_categorical(ptr(logits.contiguous()), B, K, int(seed), int(step), int(slot), ptr(probs), ptr(idx), stream()),
          'pa_sample_categorical')
    return probs, idx


def pose_step(hg, optimizer_hg, data):
    """the common tail of both branches of train_hg (:229-245 / :282-299)"""
    loss, _ = hg.loss_and_backward(img4=data['img4'], pts=data['pts'])
    optimizer_hg.step()
    pckh, _ = hg.pckh_origin_res(data['c'], data['s'], data['r'], data['grnd_pts'], data['normalizer'])
    return loss, pckh[0]


def train_hg_step(i, hg, optimizer_hg, agent_sr, augmenter, batch, seed=0):
    """one iteration of train_hg (:213-299).  Returns (kind, loss, pckh) as device scalars."""
    hg.train()
    agent_sr.eval()
    if i % 2 == 0:                                                   # regular augmentation (:224-245)
        loss, pckh = pose_step(hg, optimizer_hg, augmenter.regular(batch))
        return 'regular', loss, pckh
    std = augmenter.standard(batch)                                  # img_std (:247)
    ls, lr = hg(asn=agent_sr, img4=std['img4'], is_half_hg=True, is_aug=True)       # (:250)
    _, si = sample_bins(ls, seed, i, 0)
    _, ri = sample_bins(lr, seed, i, 1)
    loss, pckh = pose_step(hg, optimizer_hg, augmenter.agent(batch, si, ri, mode=1))     # load_batch_data(separate_s_r=False)
    return 'agent', loss, pckh


def train_hg(batches, hg, optimizer_hg, agent_sr, augmenter, epoch, opt, log=print):
    meters = {k: AverageMeter() for k in ('loss_hg_regular', 'loss_hg_sr', 'loss_hg', 'pckhs_regular', 'pckhs_sr', 'pckh')}
    n = len(batches)
    for i, batch in enumerate(batches):
        kind, loss, pckh = train_hg_step(i, hg, optimizer_hg, agent_sr, augmenter, batch, seed=epoch)
        if i % opt.print_freq == 0 or i == n - 1:
            l, p = float(loss), float(pckh)
            meters['loss_hg'].update(l); meters['pckh'].update(p)
            meters['loss_hg_regular' if kind == 'regular' else 'loss_hg_sr'].update(l)
            meters['pckhs_regular' if kind == 'regular' else 'pckhs_sr'].update(p)
            log('epoch:%d, iters:%d/%d ' % (epoch, i, n) + ' '.join('%s: %.4f' % (k, m.avg) for k, m in meters.items()))
    return meters['loss_hg'].avg, meters['pckh'].avg


def separated_s_r_pckh(hg, data_scale, data_rot):
    """compute_separated_s_r_pckh (:452-467): full forward on the scale-only and the rotation-only crops,
    per-person PCKh of each."""
    out = []
    for d in (data_scale, data_rot):
        hg(img4=d['img4'], pts=d['pts'])
        _, person = hg.pckh_origin_res(d['c'], d['s'], d['r'], d['grnd_pts'], d['normalizer'], per_person=True)
        out.append(person)
    return out


def train_agent_sr(batch, hg, agent_sr, optimizer_sr, augmenter, epoch_sr, seed=0):
    """train_agent_sr (:317-422): exactly one batch.  Returns the agent loss (device scalar)."""
    hg.eval()
    agent_sr.train()
    # regular (random) scale-only / rotation-only crops from the loader (data/joint_train_s_r_agent.py:119-158)
    check(lib().pa_sample_aug(ptr(batch.meta), None, None, 0, int(augmenter.seed), int(augmenter.step), batch.B, ptr(batch.params), stream()))
    augmenter.step += 1
    reg = batch.params.clone()
    batch.params[:, 4] = 0; batch.params[:, 5:8] = 1; batch.params[:, 0] = batch.meta[:, 0].double()      # no flip / colour in separate mode
    rot_keep = batch.params[:, 3].clone(); sc_keep = batch.params[:, 2].clone()
    batch.params[:, 3] = 0
    d_s = augmenter._finish(batch)                                   # (s_aug, r = 0)
    batch.params[:, 2] = batch.meta[:, 2].double(); batch.params[:, 3] = rot_keep
    d_r = augmenter._finish(batch)                                   # (s, r_aug)
    regular = separated_s_r_pckh(hg, d_s, d_r)
    std = augmenter.standard(batch)
    ls, lr = hg(asn=agent_sr, img4=std['img4'], is_half_hg=True, is_aug=True)       # grads only into the agent (:342)
    ps, si = sample_bins(ls, seed, epoch_sr, 0)
    pr, ri = sample_bins(lr, seed, epoch_sr, 1)
    a_s = augmenter.agent(batch, si, ri, mode=2)                     # agent scale bin, no rotation
    a_r = augmenter.agent(batch, si, ri, mode=3)                     # agent rotation bin, annotated scale
    agent = separated_s_r_pckh(hg, a_s, a_r)
    gs = util.gen_groundtruth(ps, si.long().view(-1, 1), regular[0], agent[0])     # (:380-389)
    gr = util.gen_groundtruth(pr, ri.long().view(-1, 1), regular[1], agent[1])
    # the PCKh passes in between overwrote the pose net's feature buffers the agent's backward reads: recompute the
    # same forward (pose net in eval mode, agent on batch statistics WITHOUT a second running-stat update)
    agent_sr._forward_from_pose(hg, img4=std['img4'], update_running=False)
    loss = agent_sr.loss_and_backward(gs, gr)
    optimizer_sr.step()
    del reg, sc_keep
    return loss
