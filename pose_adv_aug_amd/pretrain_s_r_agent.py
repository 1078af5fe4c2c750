"""Stage 2 of the reference on the HIP engine: collect the per-person scale / rotation "hardness" distributions
(collect-scale-ditri.py, collect-rotation-ditri.py) and pre-train the ASN agent against them (pretrain-s-r-agent.py).

For every person the trained pose net (eval mode) sees 7 deterministic crops -- scale * 2**mean_k (or rotation mean_k) --
and `1 - per_person_pckh` of the 7 crops, normalised, is the target distribution of that person
(collect-scale-ditri.py:215-240; uniform if all 7 are perfect).  The text format is the reference's: one row of 7
`%.2f` numbers per image.  Pre-training minimises K * KL(LogSoftmax(agent(half-hourglass features)) || target) for both
heads with RMSprop (pretrain-s-r-agent.py:148-205).  Everything numeric runs on the device."""
import os
from collections import OrderedDict

import numpy as np
import torch

from .data import Augmenter, DeviceBatch
from .utils.util import DeviceMeters

SCALE_MEANS = np.arange(-0.6, 0.61, 0.2)        # data/collect_scale_distri.py (scale_means): s * 2**mean
ROT_MEANS = np.arange(-60, 61, 20)              # data/collect_rotation_distri.py (rotation_means): degrees


def lost_pckh_to_distribution(lost):
    """collect-scale-ditri.py:224-238 for a [K][B] device tensor of 1 - PCKh: columns normalised to sum 1, uniform where a
    column is all zero.  Negative entries are an error in the reference (exit())."""
    if bool((lost < 0).any()):
        raise ValueError('some of tmp_pckh is negative. error...')
    tot = lost.sum(0, keepdim=True)
    uni = torch.full_like(lost, 1.0 / lost.shape[0])
    return torch.where(tot > 0, lost / tot.clamp(min=1e-30), uni).t().contiguous()          # [B][K]


def collect_data(batches, hg, augmenter, kind, save_path=None):
    """collect_data of collect-scale-ditri.py:123-255 (kind='scale') / collect-rotation-ditri.py (kind='rotation').
    Returns the list of per-image distributions (CPU tensors of 7) and appends them to `save_path` as '%.2f' rows."""
    assert kind in ('scale', 'rotation')
    means = SCALE_MEANS if kind == 'scale' else ROT_MEANS
    hg.eval()
    grnd_distri_list = []
    for i, batch in enumerate(batches):
        lost = []
        for m in means:
            d = augmenter.fixed(batch, scale_exp=float(m), rot=0.0) if kind == 'scale' else augmenter.fixed(batch, 0.0, float(m))
            hg.forward(img4=d['img4'], pts=d['pts'])
            _, person = hg.pckh_origin_res(d['c'], d['s'], d['r'], d['grnd_pts'], d['normalizer'], per_person=True)
            lost.append(1.0 - person)
        distri = lost_pckh_to_distribution(torch.stack(lost, 0)).cpu()
        for row in distri:
            grnd_distri_list.append(row.clone())
        if save_path is not None:
            with open(save_path, 'a+') as log_file:
                np.savetxt(log_file, distri.numpy(), fmt='%.2f')
    return grnd_distri_list


def read_grnd_distri_from_txt(load_path):
    """pretrain-s-r-agent.py:262-275 / collect-scale-ditri.py:257-278."""
    grnd_distri_list = []
    with open(load_path, 'r') as fd:
        for line in fd:
            tmp_vec = torch.tensor([float(x) for x in line.split()], dtype=torch.float32)
            assert abs(1 - float(tmp_vec.sum())) < 0.1
            grnd_distri_list.append(tmp_vec)
    return grnd_distri_list


def _targets(distri_list, rows, dev):
    """data/pretrain_s_r_agent.py:127-128: the distribution of DATASET index `index` (scale_distri[index]), rows
    renormalised (the text file rounds to 2 decimals)."""
    t = torch.stack([distri_list[int(r)] for r in rows], 0)
    return (t / t.sum(1, keepdim=True)).to(dev)


def _rows(batch, first):
    """dataset indices of a batch's people: the feed's own (`batch.index`, MPII feeds incl. shuffled passes) or, for
    resident synthetic batches, the position in feed order"""
    index = getattr(batch, 'index', None)
    return list(range(first, first + batch.B)) if index is None else [int(v) for v in index]


def _run(batches, scale_distri, rotation_distri, hg, agent, optimizer, augmenter, epoch, opt, log):
    losses = None
    hg.eval()
    first, n = 0, len(batches)
    for i, batch in enumerate(batches):
        dev = batch.params.device
        rows = _rows(batch, first)
        ts, tr = _targets(scale_distri, rows, dev), _targets(rotation_distri, rows, dev)
        first += batch.B
        std = augmenter.standard(batch)
        ls, lr = hg(asn=agent, img4=std['img4'], is_half_hg=True, is_aug=True)         # pretrain-s-r-agent.py:172-173
        if optimizer is not None:
            loss = agent.loss_and_backward(ts, tr, log_eps=0.0)                         # :177-190 (LogSoftmax form)
            optimizer.step()                                                            # :192-194
        else:
            loss = agent.kl_loss(ls, lr, ts, tr, log_eps=0.0)
        if losses is None:
            losses = DeviceMeters(('loss',), loss.device)
        losses.update({'loss': loss})                                                   # every iteration, no host sync
        if i % opt.print_freq == 0 or i == n - 1:
            log('%s epoch:%d, iters:%d/%d loss: %.4f' % ('sr-pretrain' if optimizer is not None else 'sr-val', epoch, i, n,
                                                          losses.averages()['loss']))
    return losses.averages()['loss']


def train(batches, scale_distri, rotation_distri, hg, agent, optimizer, augmenter, epoch, opt, log=print):
    """pretrain-s-r-agent.py:148-205."""
    agent.train()
    return _run(batches, scale_distri, rotation_distri, hg, agent, optimizer, augmenter, epoch, opt, log)


def validate(batches, scale_distri, rotation_distri, hg, agent, augmenter, epoch, opt, log=print):
    """pretrain-s-r-agent.py:207-259."""
    agent.eval()
    return _run(batches, scale_distri, rotation_distri, hg, agent, None, augmenter, epoch, opt, log)


def distribution_rows(path, people, collect):
    """The rows of one distribution file (row i = person i of the split, `people` rows).  A file that is missing or was cut short
    by an interrupted collection (collect_data appends) is collected again -- `collect(tmp_path)` writes it -- and appears under
    its name only when it is complete."""
    rows = read_grnd_distri_from_txt(path) if os.path.isfile(path) else None
    if rows is None or len(rows) != people:
        tmp = path + '.collecting'
        if os.path.isfile(tmp):
            os.remove(tmp)
        collect(tmp)
        os.replace(tmp, path)
        rows = read_grnd_distri_from_txt(path)
        if len(rows) != people:
            raise RuntimeError('%s: %d distribution rows for %d people' % (path, len(rows), people))
    return rows


def main(argv=None):
    """pretrain-s-r-agent.py:34-146: collect (if the text files are missing) + pre-train.  The agent checkpoints carry an
    ASNTrainHistory (lowest_loss / is_best by train loss, :52,:131-135) so that stage 3 loads them
    (joint-train-pose-s-r-agent.py:97-106), live in <exp>/<sr_dir>-<pose checkpoint>/ (:40-42) and resume with
    --load_prefix_sr (:64-70)."""
    from .options.train_options import TrainOptions
    from .stack_hg import make_feeds
    from .utils.checkpoint import Checkpoint
    from .utils.optim import RMSprop
    from .utils.logger import Logger
    from .utils.util import ASNTrainHistory
    from .models.asn_stacked_hg import create_hg, create_asn
    opt = TrainOptions().parse(argv)
    exp = os.path.join(opt.exp_dir, opt.exp_id)
    sr_dir = os.path.join(exp, opt.sr_dir + '-' + opt.load_prefix_pose[0:-1]) if opt.load_prefix_pose != '' else os.path.join(exp, opt.sr_dir)
    os.makedirs(sr_dir, exist_ok=True)
    hg = create_hg(num_stacks=2, num_modules=1, num_classes=16, chan=256, default_batch=opt.bs)
    if opt.load_prefix_pose != '':
        ck = Checkpoint(); ck.load_prefix = os.path.join(exp, opt.load_prefix_pose)[0:-1]
        ck.load_checkpoint(hg)
    agent = create_asn(chan_in=256, chan_out=256, scale_num=len(SCALE_MEANS), rotation_num=len(ROT_MEANS), is_aug=True,
                       default_batch=opt.bs)
    optimizer = RMSprop(agent, lr=opt.lr, alpha=0.99, eps=1e-8)                              # :93-94 (opt.lr, not agent_lr)
    history, ckpt = ASNTrainHistory(), Checkpoint()
    if opt.load_prefix_sr != '':                                                              # resume (:64-70)
        ckpt.load_prefix = sr_dir + '/' + opt.load_prefix_sr[0:-1]
        ckpt.load_checkpoint(agent, optimizer, history)
    ckpt.save_prefix = sr_dir + '/'
    aug = Augmenter(seed=4321)
    # the feeds are STREAMED every pass (an MPII split resident in HBM would be 60-130 GB of padded frames): the collection
    # pass walks each split once in dataset order (row i of a text file = person i, collect-scale-ditri.py:123-255), the
    # training passes look a person's distribution up by its dataset index (data/pretrain_s_r_agent.py:127-128)
    from .data import dataset_size
    train_feed, val_feed = make_feeds(opt)
    ordered = {}                                          # the dataset-order train feed (a second JSON parse + decoder pool): only when a file has to be collected

    def feed_of(split):
        if split == 'val':
            return val_feed
        if 'train' not in ordered:
            ordered['train'] = make_feeds(opt, shuffle_train=False, log=lambda m: None)[0]
        return ordered['train']
    distri = {}
    for split in ('train', 'val'):
        # rows of a distribution file = people of the SPLIT (the ordered feed that writes it keeps the last partial batch; the
        # shuffled training feed drops it, so ITS num_samples is (n // bs) * bs)
        people = dataset_size(train_feed if split == 'train' else val_feed)
        for kind, fname in (('scale', '%s_scales.txt' % split), ('rotation', '%s_rotations.txt' % split)):
            path = os.path.join(sr_dir, fname)
            rows = distribution_rows(path, people, lambda tmp: collect_data(feed_of(split), hg, aug, kind, tmp))
            distri[(split, kind)] = rows
    summary = sr_dir + '/' + 'training-summary.txt'
    resumed = opt.load_prefix_sr != '' and os.path.isfile(summary)
    logger = Logger(summary, title='training-summary', resume=resumed)                      # :120-122
    if not resumed:
        logger.set_names(['Epoch', 'LR', 'Train Loss', 'Val Loss'])
    start = history.epoch[-1]['epoch'] + 1 if history.epoch else 0
    for epoch in range(start, opt.nEpochs):               # (:127-141: the learning rate stays at opt.lr, adjust_lr is imported but never called)
        tl = train(train_feed, distri[('train', 'scale')], distri[('train', 'rotation')], hg, agent, optimizer, aug, epoch, opt)
        vl = validate(val_feed, distri[('val', 'scale')], distri[('val', 'rotation')], hg, agent, aug, epoch, opt)
        lr_now = optimizer.param_groups[0]['lr']
        history.update(OrderedDict([('epoch', epoch)]), OrderedDict([('lr', lr_now)]),
                       OrderedDict([('train_loss', tl), ('val_loss', vl)]))
        ckpt.save_checkpoint(agent, optimizer, history, is_asn=True)
        logger.append([epoch, lr_now, tl, vl])                                              # :141
    logger.close()

if __name__ == '__main__':
    main()
