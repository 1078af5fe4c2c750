"""utils/util.py of the reference: training histories, meters, LR schedule and the agent's reward
shaping.  Host-side bookkeeping (tiny, off the hot path); gen_groundtruth is vectorised over the batch
but follows utils/util.py:147-253 branch for branch."""
import os
from collections import OrderedDict

import torch


class PoseTrainHistory(object):
    """utils/util.py:8-46."""

    def __init__(self):
        self.epoch, self.lr, self.losses, self.pckh = [], [], [], []
        self.best_pckh = 0.
        self.is_best = True

    def update(self, epoch, lr, loss, pckh):
        self.epoch.append(epoch); self.lr.append(lr); self.losses.append(loss); self.pckh.append(pckh)
        self.is_best = pckh['val_pckh'] > self.best_pckh
        self.best_pckh = max(pckh['val_pckh'], self.best_pckh)

    def state_dict(self):
        return OrderedDict([('epoch', self.epoch), ('lr', self.lr), ('loss', self.losses), ('pckh', self.pckh),
                            ('best_pckh', self.best_pckh), ('is_best', self.is_best)])

    def load_state_dict(self, sd):
        self.epoch, self.lr, self.losses, self.pckh = sd['epoch'], sd['lr'], sd['loss'], sd['pckh']
        self.best_pckh, self.is_best = sd['best_pckh'], sd['is_best']


class ASNTrainHistory(object):
    """utils/util.py:48-86."""

    def __init__(self):
        self.epoch, self.lr, self.losses = [], [], []
        self.lowest_loss = 100.
        self.is_best = True

    def update(self, epoch, lr, loss, pckh=None):
        self.epoch.append(epoch); self.lr.append(lr); self.losses.append(loss)
        self.is_best = loss['train_loss'] < self.lowest_loss
        self.lowest_loss = min(loss['train_loss'], self.lowest_loss)

    def state_dict(self):
        return OrderedDict([('epoch', self.epoch), ('lr', self.lr), ('loss', self.losses),
                            ('lowest_loss', self.lowest_loss), ('is_best', self.is_best)])

    def load_state_dict(self, sd):
        self.epoch, self.lr, self.losses = sd['epoch'], sd['lr'], sd['loss']
        self.lowest_loss, self.is_best = sd['lowest_loss'], sd['is_best']


class AverageMeter(object):
    """utils/util.py:88-103."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class DeviceMeters(object):
    """The AverageMeters of the training loops (stack-hg.py:126-131, joint-train-pose-s-r-agent.py:197-205) for values that
    live on the GPU: every update is ONE tiny device add into a [names][2] (sum, count) tensor -- every iteration
    contributes, as in the reference, without a host synchronisation per step; .averages() reads the table back (the
    print_freq / end-of-epoch sync)."""

    def __init__(self, names, device):
        self.names = list(names)
        self.index = {n: i for i, n in enumerate(self.names)}
        self.table = torch.zeros(len(self.names), 2, dtype=torch.float64, device=device)
        self._masks = {}

    def update(self, values):
        """values: {name: 0-d device tensor}; names not given keep their sums"""
        key = tuple(sorted(values))
        m = self._masks.get(key)
        if m is None:
            m = torch.zeros(len(self.names), 2, dtype=torch.float64)
            for n in key:
                m[self.index[n], 1] = 1.0
            idx = torch.tensor([self.index[n] for n in key], dtype=torch.long)
            m = self._masks[key] = (m.to(self.table.device), idx.to(self.table.device))
        mask, idx = m
        vals = torch.stack([values[n].detach().reshape(()).double() for n in key])
        self.table.add_(mask)
        self.table[:, 0].index_add_(0, idx, vals)

    def averages(self, all_ranks=True):
        """Read the table back.  all_ranks: sum the (sum, count) tables of all data-parallel ranks first (a clone is
        reduced, the running sums stay local): the logged averages cover the GLOBAL batch, like the reference's single
        process that sees every DataParallel shard.  len(names) * 2 numbers, only at print time."""
        import torch.distributed as dist
        t = self.table
        if all_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t.cpu()
        return OrderedDict((n, float(t[i, 0] / t[i, 1]) if float(t[i, 1]) > 0 else 0.0) for i, n in enumerate(self.names))


def adjust_lr(opt, optimizer, epoch):
    """utils/util.py:105-117: x0.2 at epoch 100, x0.5 at epoch 140."""
    if epoch < 100:
        return
    if epoch == 100:
        opt.lr = opt.lr * 0.2
    elif epoch == 140:
        opt.lr = opt.lr * 0.5
    for g in optimizer.param_groups:
        g['lr'] = opt.lr


def mkdirs(paths):
    for p in (paths if isinstance(paths, list) else [paths]):
        os.makedirs(p, exist_ok=True)            # (every rank of a data-parallel run parses the options: check-then-create races)


def gen_groundtruth(pred_distri, indexes, pckh_regular, pckh_agent):
    """Reward shaping of the agent's target distribution (utils/util.py:147-253).
    pred_distri n x K probabilities, indexes n x 1 sampled bin, pckh_* n per-sample PCKh.
    If the agent's augmentation made the sample harder (pckh_regular > pckh_agent) the sampled bin is
    raised by 20 % of itself and the rest lowered uniformly, otherwise lowered by 50 % and the rest
    raised; then clamp to [0, 2/K] with redistribution and renormalise."""
    g = pred_distri.detach().clone().float().cpu()
    idx = torch.as_tensor(indexes).long().cpu()
    assert g.dim() == 2 and idx.shape == (g.size(0), 1), 'one sampled bin per row (the scale/rotation agent)'
    reg = torch.as_tensor(pckh_regular).float().cpu().reshape(-1)
    ag = torch.as_tensor(pckh_agent).float().cpu().reshape(-1)
    n, K = g.shape
    thres = (1. / K) * 2
    rows = torch.arange(n)
    sel = torch.zeros(n, K, dtype=torch.bool); sel[rows, idx[:, 0]] = True
    harder = (reg - ag) > 0
    cur = g[rows, idx[:, 0]]
    delta = torch.where(harder, 0.2 * cur, -0.5 * cur)                 # added to the sampled bin
    g[rows, idx[:, 0]] = cur + delta
    g = torch.where(sel, g, g - (delta / (K - 1))[:, None])
    # clamp pass, in index order like the reference loop (:209-236)
    over = (g > thres); under = (g < 0) & ~over
    over_part = torch.where(over, g - thres, torch.zeros_like(g)).sum(1)
    under_part = torch.where(under, g, torch.zeros_like(g)).sum(1)
    g = torch.where(over, torch.full_like(g, thres), g)
    g = torch.where(under, torch.zeros_like(g), g)
    gap = over_part + under_part
    n_not_over = (~over).sum(1).clamp_min(1).float()
    n_not_under = (~under).sum(1).clamp_min(1).float()
    pos = gap > 0
    neg = gap < 0
    g = torch.where(pos[:, None] & ~over, g + (gap / n_not_over)[:, None], g)
    g2 = torch.where(neg[:, None] & ~under, g + (gap / n_not_under)[:, None], g)
    g = torch.where(neg[:, None] & ~under, g2.clamp_min(0), g)
    g = g / g.sum(1, keepdim=True)
    return g.to(pred_distri.device)
