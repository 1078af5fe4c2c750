"""joint-train-pose-s-r-agent.py of the reference (stage 3: adversarial scale/rotation augmentation) on the
HIP engine.  The DataLoader that the reference re-forks for every agent batch (load_batch_data, :425-450)
becomes a re-crop of the resident frames with the sampled bins: the agent's augmentations never touch the host.

  main            :38-193    joint_dir = <exp>/<joint_dir>-<pose checkpoint>, pose-* / agent-* checkpoints, per epoch:
                             train_hg -> validate -> checkpoint -> ONE train_agent_sr batch -> checkpoint
  train_hg        :195-315   even steps: regular augmentation; odd steps: half-hourglass (pose net in TRAIN
                             mode, Appendix A.9) + agent (eval) -> softmax -> categorical draw per sample ->
                             agent-law crop -> full pose step
  train_agent_sr  :317-422   pose net in eval mode, agent in train mode; per-person PCKh of regular
                             scale-only / rotation-only crops vs the agent's; reward-shaped targets
                             (utils/util.gen_groundtruth); KL loss; ONE batch per call (Appendix A.8)
"""
import os
from collections import OrderedDict

import torch

from ._lib import lib, check, ptr, stream
from .stack_hg import PCK_IDX
from .utils import util
from .utils.util import DeviceMeters

METER_NAMES = ('loss_hg_regular', 'loss_hg_sr', 'loss_hg', 'pckhs_regular', 'pckhs_sr', 'pckh')      # :197-205, :300-305


def _bin_seed(seed, augmenter):
    """the categorical draws of a rank come from ITS OWN stream (the augmenter's seed carries the rank): with one common
    stream every rank would reuse the same uniforms for its shard, and the global batch's draws would not be independent
    like the np.random.choice calls of the reference's single process (:252-271)"""
    return (int(seed) * 1000003 + int(augmenter.seed)) & 0x7fffffffffffffff


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
    _, si = sample_bins(ls, _bin_seed(seed, augmenter), i, 0)
    _, ri = sample_bins(lr, _bin_seed(seed, augmenter), i, 1)
    loss, pckh = pose_step(hg, optimizer_hg, augmenter.agent(batch, si, ri, mode=1))     # load_batch_data(separate_s_r=False)
    return 'agent', loss, pckh


def train_hg(batches, hg, optimizer_hg, agent_sr, augmenter, epoch, opt, log=print):
    """:195-315.  All six meters see every iteration (device-side sums, split by the kind of step); the host synchronises
    every print_freq steps only."""
    meters = None
    n = len(batches)
    for i, batch in enumerate(batches):
        kind, loss, pckh = train_hg_step(i, hg, optimizer_hg, agent_sr, augmenter, batch, seed=epoch)
        if meters is None:
            meters = DeviceMeters(METER_NAMES, loss.device)
        tag = ('loss_hg_regular', 'pckhs_regular') if kind == 'regular' else ('loss_hg_sr', 'pckhs_sr')
        meters.update({'loss_hg': loss, 'pckh': pckh, tag[0]: loss, tag[1]: pckh})
        if i % opt.print_freq == 0 or i == n - 1:
            log('epoch:%d, iters:%d/%d ' % (epoch, i, n) + ' '.join('%s: %.4f' % kv for kv in meters.averages().items()))
            from .stack_hg import warn_skipped_steps
            warn_skipped_steps(optimizer_hg, log)         # (fp16 build: a skipped step must not pass unnoticed)
    if meters is None:                                    # an empty feed
        return 0.0, 0.0
    d = meters.averages()
    return d['loss_hg'], d['pckh']


def separated_s_r_pckh(hg, data_scale, data_rot):
    """compute_separated_s_r_pckh (:452-467): full forward on the scale-only and the rotation-only crops,
    per-person PCKh of each."""
    out = []
    for d in (data_scale, data_rot):
        hg(img4=d['img4'], pts=d['pts'])
        _, person = hg.pckh_origin_res(d['c'], d['s'], d['r'], d['grnd_pts'], d['normalizer'], per_person=True)
        out.append(person)
    return out


def regular_separated_crops(batch, augmenter, want_nchw=False):
    """what the AGENT loader hands over with separate_s_r (data/joint_train_s_r_agent.py:134-160): the regular law's scale
    with no rotation, and its rotation at the annotated scale; no flip, no colour jitter in this mode."""
    check(lib().pa_sample_aug(ptr(batch.meta), None, None, 0, int(augmenter.seed), int(augmenter.step), batch.B, ptr(batch.params), stream()))
    augmenter.step += 1
    p = batch.params
    p[:, 4] = 0; p[:, 5:8] = 1; p[:, 0] = batch.meta[:, 0].double()
    rot_keep = p[:, 3].clone()
    p[:, 3] = 0
    d_s = augmenter._finish(batch, want_nchw)                        # (s_aug, r = 0)
    p[:, 2] = batch.meta[:, 2].double(); p[:, 3] = rot_keep
    d_r = augmenter._finish(batch, want_nchw)                        # (s, r_aug)
    return d_s, d_r


def train_agent_sr(batch, hg, agent_sr, optimizer_sr, augmenter, epoch_sr, seed=0, trace=None, pckh_override=None):
    """train_agent_sr (:317-422): exactly one batch.  Returns the agent loss (device scalar).
    trace: a dict that receives every intermediate (crops as fp32 NCHW, joints, bins, PCKh lists, targets) -- the parity
    test replays them through the oracle's restatement of the same function; pckh_override = ([reg_s, reg_r], [agent_s,
    agent_r]) replaces the four per-person PCKh vectors there (an untrained net scores 0 everywhere)."""
    hg.eval()
    agent_sr.train()
    want = trace is not None
    d_s, d_r = regular_separated_crops(batch, augmenter, want)
    regular = separated_s_r_pckh(hg, d_s, d_r)                       # :331-336
    std = augmenter.standard(batch, want)
    ls, lr = hg(asn=agent_sr, img4=std['img4'], is_half_hg=True, is_aug=True)       # grads only into the agent (:342)
    ps, si = sample_bins(ls, _bin_seed(seed, augmenter), epoch_sr, 0)
    pr, ri = sample_bins(lr, _bin_seed(seed, augmenter), epoch_sr, 1)
    a_s = augmenter.agent(batch, si, ri, mode=2, want_nchw=want)     # agent scale bin, no rotation
    a_r = augmenter.agent(batch, si, ri, mode=3, want_nchw=want)     # agent rotation bin, annotated scale
    agent = separated_s_r_pckh(hg, a_s, a_r)                         # :372-374
    if pckh_override is not None:
        regular, agent = pckh_override
    gs = util.gen_groundtruth(ps, si.long().view(-1, 1), regular[0], agent[0])     # (:380-389)
    gr = util.gen_groundtruth(pr, ri.long().view(-1, 1), regular[1], agent[1])
    # the PCKh passes in between overwrote the pose net's feature buffers the agent's backward reads: recompute the
    # same forward (pose net in eval mode, agent on batch statistics WITHOUT a second running-stat update)
    agent_sr._forward_from_pose(hg, img4=std['img4'], update_running=False)
    loss = agent_sr.loss_and_backward(gs, gr)
    optimizer_sr.step()
    if want:
        trace.update(regular=[d_s, d_r], agent=[a_s, a_r], std=std, logits=(ls, lr), probs=(ps, pr), bins=(si, ri),
                     pckh_regular=regular, pckh_agent=agent, targets=(gs, gr), loss=loss)
    return loss


def main(argv=None):
    """joint-train-pose-s-r-agent.py:38-193 (directory layout, checkpoint prefixes, per-epoch order, both histories)."""
    from .data import Augmenter
    from .models.asn_stacked_hg import create_asn, create_hg
    from .options.train_options import TrainOptions
    from .pretrain_s_r_agent import ROT_MEANS, SCALE_MEANS
    from .stack_hg import broadcast_parameters, init_distributed, make_feeds, validate
    from .utils.checkpoint import Checkpoint
    from .utils.logger import Logger
    from .utils.optim import RMSprop
    from .utils.util import ASNTrainHistory, PoseTrainHistory, adjust_lr
    from .utils.visualizer import Visualizer
    opt = TrainOptions().parse(argv)
    if opt.joint_dir == '':
        print('joint directory is null.')
        raise SystemExit(1)
    if opt.load_prefix_pose == '':
        print('please input the checkpoint name of the pose model')
        raise SystemExit(1)
    if opt.load_prefix_sr == '':
        print('please input the checkpoint name of the sr agent.')
        raise SystemExit(1)
    rank, world, _ = init_distributed()
    exp = os.path.join(opt.exp_dir, opt.exp_id)
    joint_dir = os.path.join(exp, opt.joint_dir + '-' + opt.load_prefix_pose[0:-1])          # :44-46
    if rank == 0 and not os.path.isdir(joint_dir):
        os.makedirs(joint_dir)
    vis = Visualizer(opt, log_path=(joint_dir + '/' + ('train-log.txt' if opt.is_train else 'val-log.txt')) if rank == 0 else None)
    log = (lambda m: (print(m), vis.write_log(m))) if rank == 0 else (lambda m: None)

    hg = create_hg(num_stacks=2, num_modules=1, num_classes=16, chan=256, default_batch=opt.bs)          # :61-62
    optimizer_hg = RMSprop(hg, lr=opt.lr, alpha=0.99, eps=1e-8)
    train_history_pose, checkpoint_hg = PoseTrainHistory(), Checkpoint()
    if opt.load_checkpoint:                                                                   # resume the joint stage (:72-74)
        checkpoint_hg.load_prefix = joint_dir + '/' + opt.load_prefix_pose[0:-1]
        ok = checkpoint_hg.load_checkpoint(hg, optimizer_hg, train_history_pose)
    else:                                                                                     # start from stage 1 (:75-80)
        checkpoint_hg.load_prefix = exp + '/' + opt.load_prefix_pose[0:-1]
        ok = checkpoint_hg.load_checkpoint(hg, optimizer_hg, train_history_pose)
        for g in optimizer_hg.param_groups:
            g['lr'] = opt.lr
    if not ok:
        raise SystemExit('pose checkpoint %s.pth.tar not found' % checkpoint_hg.load_prefix)
    checkpoint_hg.save_prefix = joint_dir + '/pose-'                                          # :81

    agent_sr = create_asn(chan_in=256, chan_out=256, scale_num=len(SCALE_MEANS), rotation_num=len(ROT_MEANS), is_aug=True,
                          default_batch=opt.bs)                                               # :87-90
    optimizer_sr = RMSprop(agent_sr, lr=opt.agent_lr, alpha=0.99, eps=1e-8)
    train_history_sr, checkpoint_sr = ASNTrainHistory(), Checkpoint()
    if opt.load_checkpoint:                                                                   # :99-101
        checkpoint_sr.load_prefix = joint_dir + '/' + opt.load_prefix_sr[0:-1]
        ok = checkpoint_sr.load_checkpoint(agent_sr, optimizer_sr, train_history_sr)
    else:                                                                                     # :102-109
        sr_pretrain_dir = os.path.join(exp, opt.sr_dir + '-' + opt.load_prefix_pose[0:-1])
        checkpoint_sr.load_prefix = sr_pretrain_dir + '/' + opt.load_prefix_sr[0:-1]
        ok = checkpoint_sr.load_checkpoint(agent_sr, optimizer_sr, train_history_sr)
        for g in optimizer_sr.param_groups:
            g['lr'] = opt.agent_lr
    if not ok:
        raise SystemExit('agent checkpoint %s.pth.tar not found' % checkpoint_sr.load_prefix)
    checkpoint_sr.save_prefix = joint_dir + '/agent-'                                         # :110
    broadcast_parameters(hg)
    broadcast_parameters(agent_sr)

    augmenter = Augmenter(seed=4242 + rank)
    train_feed, val_feed = make_feeds(opt, rank, world, log)                                  # :117-133
    if not opt.is_train:                                                                      # :137-144
        _, _, predictions = validate(val_feed, hg, augmenter, train_history_pose.epoch[-1]['epoch'], opt, log=log)
        if rank == 0:
            checkpoint_hg.save_preds(predictions)
        return
    logger = None
    if rank == 0:
        logger = Logger(joint_dir + '/' + 'pose-training-summary.txt', title='pose-training-summary')     # :145-147
        logger.set_names(['Epoch', 'LR', 'Train Loss', 'Val Loss', 'Train PCKh', 'Val PCKh'])
    start_epoch_pose = train_history_pose.epoch[-1]['epoch'] + 1                              # :149-150
    epoch_sr = train_history_sr.epoch[-1]['epoch'] + 1
    for epoch in range(start_epoch_pose, opt.nEpochs):
        adjust_lr(opt, optimizer_hg, epoch)
        train_loss_pose, train_pckh = train_hg(train_feed, hg, optimizer_hg, agent_sr, augmenter, epoch, opt, log=log)
        val_loss, val_pckh, predictions = validate(val_feed, hg, augmenter, epoch, opt, log=log)
        lr_now = optimizer_hg.param_groups[0]['lr']
        train_history_pose.update(OrderedDict([('epoch', epoch)]), OrderedDict([('lr', lr_now)]),
                                  OrderedDict([('train_loss', train_loss_pose), ('val_loss', val_loss)]),
                                  OrderedDict([('train_pckh', train_pckh), ('val_pckh', val_pckh)]))
        if rank == 0:
            checkpoint_hg.save_checkpoint(hg, optimizer_hg, train_history_pose, predictions)  # :171-173
            logger.append([epoch, lr_now, train_loss_pose, val_loss, train_pckh, val_pckh])
        # one agent batch per epoch (:180-191, Appendix A.8)
        batch = next(iter(train_feed))
        train_loss_sr = float(train_agent_sr(batch, hg, agent_sr, optimizer_sr, augmenter, epoch_sr, seed=epoch))
        log('epoch:%d, iters:0/1 loss_agent_sr: %.4f ' % (epoch_sr, train_loss_sr))
        from .stack_hg import warn_skipped_steps
        warn_skipped_steps(optimizer_sr, log, 'agent')        # (fp16 build; the agent optimizer has its own skip state)
        train_history_sr.update(OrderedDict([('epoch', epoch_sr)]), OrderedDict([('lr', optimizer_sr.param_groups[0]['lr'])]),
                                OrderedDict([('train_loss', train_loss_sr), ('val_loss', 0)]))
        if rank == 0:
            checkpoint_sr.save_checkpoint(agent_sr, optimizer_sr, train_history_sr, is_asn=True)      # :187-189
        epoch_sr += 1
    if logger is not None:
        logger.close()


if __name__ == '__main__':
    main()
