"""stack-hg.py of the reference (stage-1 pose training) on the HIP engine.

The per-batch body of train() (stack-hg.py:134-185) becomes `train_step`: augmentation (regular law) and
warp on the device, forward + Gaussian-target MSE + hand-written backward, ONE RCCL all-reduce of the
flat gradient (inside optimizer.step), fused RMSprop, PCKh in heat-map space and at original resolution
on the device -- no host synchronisation inside the step.  One process per GPU (torchrun)."""
import os
from collections import OrderedDict

import torch
import torch.distributed as dist

from .data import AugmentAhead, Augmenter, BatchFeed, DeviceBatch, num_samples, with_next
from .models.asn_stacked_hg import create_hg
from .utils.optim import RMSprop
from .utils.util import AverageMeter, DeviceMeters, PoseTrainHistory, adjust_lr

PCK_IDX = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]            # stack-hg.py:89


def init_distributed():
    """One process per GPU; backend 'nccl' is RCCL on ROCm.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    # POSEADV_FORCE_DIST=1: take the distributed branch with WORLD_SIZE=1 too (one rank through RCCL: the collective and
    # the overlapped bucket path run on the engine's real streams on a single-GPU box, tests/test_gpu_rccl.py)
    force = os.environ.get('POSEADV_FORCE_DIST') == '1'
    if (world > 1 or force) and not dist.is_initialized():
        backend = os.environ.get('POSEADV_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')     # nccl = RCCL over xGMI
        init = os.environ.get('POSEADV_DIST_INIT')               # e.g. file:///tmp/rendezvous (no TCP port at all); default: env://
        if init:
            dist.init_process_group(backend, init_method=init, rank=rank, world_size=world)
        else:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def dist_active():
    """True when gradients have to go through the process group: world > 1, or a forced single-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('POSEADV_FORCE_DIST') == '1')


def broadcast_parameters(net):
    """Identical replicas at start (replaces DataParallel's per-forward broadcast, stack-hg.py:49)."""
    if dist_active():
        net._ensure_table()
        dist.broadcast(net.flat_params, src=0)
        dist.broadcast(net.flat_buffers, src=0)
        net.weights_changed()


def train_step(net, optimizer, augmenter, batch, want_pckh=True, data=None):
    """stack-hg.py:134-180 for one batch.  Returns device scalars (loss, pckh, pckh_origin_res).
    data: the batch's augmented input if it was prepared ahead (data.AugmentAhead), else it is made here."""
    if data is None:
        data = augmenter.regular(batch)
    def meters():
        pckh = net.accuracy(PCK_IDX)                                                    # stack-hg.py:176
        pckh_o, _ = net.pckh_origin_res(data['c'], data['s'], data['r'], data['grnd_pts'], data['normalizer'])   # :178
        return pckh, pckh_o
    if want_pckh and not getattr(net, 'use_graph', False):
        # the meters read the forward pass's heat maps only: launched between the two passes they run beside the backward pass on the
        # engine's meter stream (100 us of short launches that otherwise sit between two steps); same values
        loss, _ = net.loss_and_backward(img4=data['img4'], pts=data['pts'], after_forward=meters)
        optimizer.step()
        pckh, pckh_o = net.after_forward_result
        return loss, pckh[0], pckh_o[0]
    loss, _ = net.loss_and_backward(img4=data['img4'], pts=data['pts'])
    optimizer.step()
    if not want_pckh:
        return loss, None, None
    pckh, pckh_o = meters()
    return loss, pckh[0], pckh_o[0]


import weakref
_SKIPPED_SEEN = weakref.WeakKeyDictionary()      # optimizer -> count already reported


def warn_skipped_steps(optimizer, log=print, name=''):
    """fp16 build: the engine skips (and counts) an optimizer step whose scaled gradient holds inf / NaN.  Silent skipping would hide a
    persistent overflow, so the loops report every growth of the counter where they synchronise anyway (at print time).  The counter
    belongs to the optimizer (pa_rmsprop_step_state: its own {flag, skipped} device pair), so what was reported is tracked per optimizer
    too: the joint loop asks for the pose net's and for the agent's."""
    from . import _lib
    if _lib.DTYPE != 'fp16' or not hasattr(optimizer, 'skipped_steps'):
        return 0
    n = optimizer.skipped_steps()
    seen = _SKIPPED_SEEN.get(optimizer, 0)
    if n > seen:
        log('WARNING: %d %soptimizer step(s) skipped for a non-finite fp16 gradient (%d so far): the gradient scale PA_GRAD_SCALE '
            'overflows this model -- train in the bf16 build or lower the scale' % (n - seen, name and name + ' ', n))
        _SKIPPED_SEEN[optimizer] = n
    return n - seen


def train(batches, net, optimizer, augmenter, epoch, opt, log=print):
    """stack-hg.py:124-189 over a sized feed of DeviceBatches (a list, data.BatchFeed, MPII.batches()).  The meters see
    EVERY iteration (:171-180) through device-side sums; the host synchronises only every print_freq steps."""
    net.train()
    n = len(batches)
    meters = None
    ahead = AugmentAhead(augmenter)
    for i, (batch, nxt) in enumerate(with_next(batches)):
        if i == 0:
            ahead.start(batch)
        data = ahead.take()
        ahead.start(nxt)                                   # the next batch's crop runs beside this step's forward pass
        loss, pckh, pckh_o = train_step(net, optimizer, augmenter, batch, data=data)
        if meters is None:
            meters = DeviceMeters(('loss', 'pckh', 'pckh_origin_res'), loss.device)
        meters.update({'loss': loss, 'pckh': pckh, 'pckh_origin_res': pckh_o})
        if i % opt.print_freq == 0 or i == n - 1:          # the only host sync
            log('epoch:%d, iters:%d/%d ' % (epoch, i, n) + ''.join('%s: %.4f ' % kv for kv in meters.averages().items()))   # utils/visualizer.py:70-72
            warn_skipped_steps(optimizer, log)
    if meters is None:                                    # an empty feed
        return 0.0, 0.0
    d = meters.averages()
    return d['loss'], d['pckh_origin_res']


def validate_step(net, augmenter, batch):
    """stack-hg.py:206-258 for one batch, on the device: forward on the un-augmented crop, loss against the
    Gaussian target, second forward on the W-mirrored input, flip back + left/right channel swap + average
    (flip test-time augmentation), PCKh in heat-map space and at the original resolution, final predictions.
    Returns (loss, pckh, pckh_origin_res, preds [B][16][2], merged heat maps)."""
    from .pylib import Evaluation, HumanAug, HumanPts
    data = augmenter.standard(batch)
    out1 = net.forward(img4=data['img4'], pts=data['pts'])                    # stack-hg.py:215-219 (loss inside)
    loss = net._last_losses.sum()
    out2 = net.forward(img4=HumanAug.flip_lr_img4(data['img4']))              # :222-226
    output = HumanAug.flip_tta_merge(out1[-1], out2[-1])                      # :227-229
    target = HumanPts.pts2heatmap_batch(data['pts'], output.shape[2], output.shape[3])
    pckh = Evaluation.accuracy(output, target, PCK_IDX)                       # :235
    res = [output.shape[2], output.shape[3]]
    pckh_o = Evaluation.accuracy_origin_res(output, data['c'], data['s'], res, data['grnd_pts'], data['normalizer'], data['r'])
    preds = Evaluation.final_preds(output, data['c'], data['s'], res, data['r'])     # :253
    return loss, pckh[0], pckh_o[0], preds, output


def validate(batches, net, augmenter, epoch, opt, num_classes=16, log=print):
    """stack-hg.py:191-260: returns (losses.avg, pckhs_origin_res.avg, predictions [N][num_classes][2])."""
    losses, pckhs, pckhs_o = AverageMeter(), AverageMeter(), AverageMeter()
    n_total = num_samples(batches)                    # (a feed is not consumed by asking for its size)
    predictions = torch.zeros(n_total, num_classes, 2)
    net.eval()
    n = len(batches)
    off = 0
    for i, batch in enumerate(batches):
        loss, pckh, pckh_o, preds, _ = validate_step(net, augmenter, batch)
        losses.update(float(loss)); pckhs.update(float(pckh)); pckhs_o.update(float(pckh_o))
        index = getattr(batch, 'index', None)
        index = list(range(off, off + batch.B)) if index is None else [int(v) for v in index]
        predictions[index] = preds.cpu()                                      # :254-255
        off += batch.B
        if i % opt.print_freq == 0 or i == n - 1:
            d = OrderedDict([('loss', losses.avg), ('pckh', pckhs.avg), ('pckh_origin_res', pckhs_o.avg)])
            log('epoch:%d, iters:%d/%d ' % (epoch, i, n) + ''.join('%s: %.4f ' % kv for kv in d.items()))
    return losses.avg, pckhs_o.avg, predictions


MPII_JSON = 'mpii-hr-lsp-normalizer.json'                # stack-hg.py:73 ('dataset/mpii-hr-lsp-normalizer.json')


def make_feeds(opt, rank=0, world=1, log=print, shuffle_train=True):
    """stack-hg.py:73-83: the MPII train / validation loaders.  With <data_dir>/mpii-hr-lsp-normalizer.json present the
    feeds are MPII.batches() (JSON + image decode on the host, everything else on the device; rank r trains on batches
    r, r + world, ... of one common order); otherwise synthetic MPII-shape people resident in HBM (the benchmark input)."""
    path = os.path.join(opt.data_dir, MPII_JSON)
    if os.path.isfile(path):
        from .mpii_for_mpii import MPII
        tr = MPII(path, opt.data_dir, is_train=True, log=log)
        va = MPII(path, opt.data_dir, is_train=False, log=log)
        return (tr.batches(opt.bs, shuffle=shuffle_train, seed=0, drop_last=shuffle_train, workers=opt.nThreads, rank=rank, world=world),
                va.batches(opt.bs, shuffle=False, workers=opt.nThreads))
    log('no %s: synthetic MPII-shape people' % path)
    return (BatchFeed.of(DeviceBatch.synthetic(opt.bs, seed=rank * 1000 + k) for k in range(4)),
            BatchFeed.of(DeviceBatch.synthetic(opt.bs, seed=500000 + k) for k in range(2)))


def main(argv=None):
    from .options.train_options import TrainOptions
    from .utils.checkpoint import Checkpoint
    from .utils.visualizer import Visualizer
    opt = TrainOptions().parse(argv)
    rank, world, _ = init_distributed()
    net = create_hg(num_stacks=2, num_modules=1, num_classes=16, chan=256, default_batch=opt.bs)     # stack-hg.py:40-41
    optimizer = RMSprop(net, lr=opt.lr, alpha=0.99, eps=1e-8)
    history, ckpt = PoseTrainHistory(), Checkpoint()
    exp_dir = os.path.join(opt.exp_dir, opt.exp_id)
    if opt.load_prefix_pose != '':
        ckpt.save_prefix = os.path.join(exp_dir, opt.load_prefix_pose)
        ckpt.load_prefix = os.path.join(exp_dir, opt.load_prefix_pose)[0:-1]
        ckpt.load_checkpoint(net, optimizer, history)
    else:
        ckpt.save_prefix = exp_dir + '/'
    broadcast_parameters(net)
    augmenter = Augmenter(seed=1234 + rank)
    vis = Visualizer(opt, log_path=os.path.join(exp_dir, 'train-log.txt' if opt.is_train else 'val-log.txt') if rank == 0 else None)
    log = (lambda m: (print(m), vis.write_log(m))) if rank == 0 else (lambda m: None)
    batches, val_batches = make_feeds(opt, rank, world, log)
    if not opt.is_train:                                                                     # stack-hg.py:90-96
        epoch = history.epoch[-1]['epoch'] if history.epoch else 0
        _, _, predictions = validate(val_batches, net, augmenter, epoch, opt, log=log)
        if rank == 0:
            ckpt.save_preds(predictions)
        return
    start = history.epoch[-1]['epoch'] + 1 if history.epoch else 0
    for epoch in range(start, opt.nEpochs):
        adjust_lr(opt, optimizer, epoch)
        tl, tp = train(batches, net, optimizer, augmenter, epoch, opt, log=log)
        vl, vp, predictions = validate(val_batches, net, augmenter, epoch, opt, log=log)      # stack-hg.py:98-99
        history.update(OrderedDict([('epoch', epoch)]), OrderedDict([('lr', optimizer.param_groups[0]['lr'])]),
                       OrderedDict([('train_loss', tl), ('val_loss', vl)]), OrderedDict([('train_pckh', tp), ('val_pckh', vp)]))
        if rank == 0:
            ckpt.save_checkpoint(net, optimizer, history, predictions)                        # :108 (+ -preds.mat)


if __name__ == '__main__':
    main()
