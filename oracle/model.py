"""Oracle (test infrastructure, see oracle/__init__.py): stacked hourglass and
ASN scale/rotation agent as plain PyTorch-CPU fp32 modules.

Parameter / buffer NAMES, SHAPES and registration ORDER are those of the
reference module tree (models/asn_stacked_hg.py), so a reference checkpoint
loads key-for-key and optimizer state indices line up.  The restatement is
table-driven rather than hand-unrolled.

  Residual        models/asn_stacked_hg.py:11-49   (post-activation bottleneck,
                                                     add BEFORE bn3, all convs biased)
  Hourglass       models/asn_stacked_hg.py:51-213  (4 levels, num_modules residuals per site)
  HourglassNet    models/asn_stacked_hg.py:215-342 (stem + stacks + re-injection)
  reference init  models/asn_stacked_hg.py:258-270 (conv U(+-1/sqrt(k*k*Cin)), BN gamma U(0,1))
  ASN             models/asn_stacked_hg.py:349-439 (scale/rotation head and occlusion-mask head)
  occlusion       models/asn_stacked_hg.py:79-136,172-190 (_dropout, _sample_mask, routing :308-321)
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class Residual(nn.Module):
    # models/asn_stacked_hg.py:13-28 ctor, :30-49 forward
    def __init__(self, cin, cout, with_adapter=False):
        super().__init__()
        mid = cout // 2                      # py2 integer division in the reference (:17)
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=True)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, padding=1, bias=True)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=True)
        self.bn3 = nn.BatchNorm2d(cout)
        # the reference registers `adapter` last (:28) -> it is last in state_dict order
        self.adapter = nn.Conv2d(cin, cout, 1, bias=True) if with_adapter else None

    def forward(self, x):
        sc = x if self.adapter is None else self.adapter(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.conv3(y) + sc               # add happens before bn3 (:44-47)
        return F.relu(self.bn3(y))


def _res_stack(chan, n):
    return nn.Sequential(*[Residual(chan, chan) for _ in range(n)])


_HG_SITES = ('down1', 'down2', 'down3', 'down4', 'up1', 'up2', 'up3', 'up4',
             'skip1', 'skip2', 'skip3', 'skip4', 'neck')   # registration order (:56-68)


class Hourglass(nn.Module):
    def __init__(self, chan, num_modules):
        super().__init__()
        for site in _HG_SITES:
            setattr(self, site, _res_stack(chan, num_modules))

    def encode(self, x):
        """Down path (:140-157).  Returns neck output and the 4 skip tensors."""
        skips = []
        for lvl in (1, 2, 3, 4):
            skips.append(getattr(self, 'skip%d' % lvl)(x))
            x = F.max_pool2d(x, 2, 2)
            x = getattr(self, 'down%d' % lvl)(x)
        return self.neck(x), skips

    def decode(self, x, skips):
        """Up path (:192-203): residual, nearest x2, add skip."""
        for lvl in (4, 3, 2, 1):
            x = getattr(self, 'up%d' % lvl)(x)
            x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = x + skips[lvl - 1]
        return x

    def forward(self, x, dropout_masks=None):
        neck, skips = self.encode(x)
        if dropout_masks is not None:        # :183-189: later stacks reuse the masks sampled in stack 0
            neck, skips = self.dropout_all(neck, skips, dropout_masks)
        return self.decode(neck, skips)

    @staticmethod
    def dropout(x, masks):
        """_dropout (:79-100): the n x 1 x 4 x 4 cell mask, nearest-upsampled to the map, multiplies every channel."""
        scale = x.shape[2] // 4              # py2 integer division (:86)
        if scale != 1:
            masks = F.interpolate(masks, scale_factor=scale, mode='nearest')
        return x * masks.expand(x.size())

    def dropout_all(self, neck, skips, masks):
        # :174-178 / :184-188: the neck and the four skip tensors, not the down path
        return self.dropout(neck, masks), [self.dropout(s, masks) for s in skips]

    def agent_features(self, x):
        """Detached feature dict handed to the agent (:159-164)."""
        neck, skips = self.encode(x)
        feats = {'neck': neck.detach()}
        for i, s in enumerate(skips):
            feats['skip%d' % (i + 1)] = s.detach()
        return feats, neck, skips


def reference_init_(module):
    """models/asn_stacked_hg.py:258-270 (same rule in ASN :380-392)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.in_channels
            b = 1.0 / math.sqrt(n)
            m.weight.data.uniform_(-b, b)
            if m.bias is not None:
                m.bias.data.uniform_(-b, b)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_()
            m.bias.data.zero_()


class HourglassNet(nn.Module):
    def __init__(self, num_modules, num_stacks, chan=256, num_classes=16):
        super().__init__()
        self.num_stacks = num_stacks
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=True)
        self.bn1 = nn.BatchNorm2d(64)
        self.residual1 = Residual(64, 128, with_adapter=True)
        self.residual2 = Residual(128, 128)
        self.residual3 = Residual(128, chan, with_adapter=True)
        self.hg = nn.ModuleList(Hourglass(chan, num_modules) for _ in range(num_stacks))
        self.post_res = nn.ModuleList(_res_stack(chan, num_modules) for _ in range(num_stacks))
        self.linear = nn.ModuleList(
            nn.Sequential(nn.Conv2d(chan, chan, 1, bias=True), nn.BatchNorm2d(chan), nn.ReLU(inplace=True))
            for _ in range(num_stacks))
        self.out_conv = nn.ModuleList(nn.Conv2d(chan, num_classes, 1, bias=True) for _ in range(num_stacks))
        self.forth_conv = nn.ModuleList(nn.Conv2d(chan, chan, 1, bias=True) for _ in range(num_stacks - 1))
        self.in_conv = nn.ModuleList(nn.Conv2d(num_classes, chan, 1, bias=True) for _ in range(num_stacks - 1))
        reference_init_(self)

    def stem(self, x):
        # :283-289
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.residual1(x)
        x = F.max_pool2d(x, 2, 2)
        x = self.residual2(x)
        return self.residual3(x)

    def forward(self, x, asn=None, is_half_hg=False, is_aug=False, is_dropout=False, choice=None):
        """:282-342.  Returns list of per-stack heatmaps; with an agent and
        is_half_hg the two agent logit tensors (:300-304).  Occlusion branch (is_dropout, :308-321):
        the agent's 4x4 mask logits; with the whole hourglass two cells per sample are drawn from their
        softmax (`choice`: np.random.choice-compatible callable) and zeroed in the neck / skip tensors of
        EVERY stack; returns (outs, pred_mask, indexes)."""
        x = self.stem(x)
        outs = []
        logits = None
        dropout_masks = indexes = None
        for i in range(self.num_stacks):
            if i == 0 and asn is not None:
                assert is_aug != is_dropout
                feats, neck, skips = self.hg[0].agent_features(x)
                if is_aug:
                    logits = asn(feats, is_aug=True)
                    if is_half_hg:
                        return logits
                else:
                    logits = asn(feats, is_dropout=True)
                    if is_half_hg:
                        return logits
                    dropout_masks, indexes = sample_mask(logits, choice)
                    neck, skips = self.hg[0].dropout_all(neck, skips, dropout_masks)
                y = self.hg[0].decode(neck, skips)
            elif dropout_masks is not None:
                y = self.hg[i](x, dropout_masks=dropout_masks)
            else:
                y = self.hg[i](x)
            y = self.linear[i](self.post_res[i](y))
            heat = self.out_conv[i](y)
            outs.append(heat)
            if i < self.num_stacks - 1:
                x = x + self.forth_conv[i](y) + self.in_conv[i](heat)   # :331-334
        if asn is not None:
            if is_dropout:
                return outs, logits, indexes
            return outs, logits[0], logits[1]
        return outs


def sample_mask(pred_masks, choice=None, dropout_num=2):
    """_Hourglass._sample_mask (:102-136): softmax over the H*W cells of each sample's n x 1 x H x W mask
    logits, `dropout_num` DISTINCT cells drawn with those probabilities (np.random.choice(replace=False)),
    mask = 1 everywhere except the drawn cells.  Returns (masks n x 1 x H x W, indexes n x dropout_num int64)."""
    import numpy as np
    choice = choice or np.random.choice
    n, c, h, w = pred_masks.shape
    assert c == 1 and h == w
    probs = torch.softmax(pred_masks.reshape(n, -1), 1).detach().cpu().numpy()
    masks = torch.ones(pred_masks.shape)
    indexes = torch.zeros(n, dropout_num, dtype=torch.long)
    for i in range(n):
        picked = choice(h * w, dropout_num, p=probs[i], replace=False)
        for j, cell in enumerate(picked):
            masks[i, 0, int(cell) // w, int(cell) % w] = 0
            indexes[i, j] = int(cell)
    return masks, indexes


def masks_from_indexes(indexes, n_cells_side=4):
    """the mask tensor _sample_mask builds for given cell indexes (n x k) -- used to replay a stored draw"""
    n = indexes.shape[0]
    masks = torch.ones(n, 1, n_cells_side, n_cells_side)
    for i in range(n):
        for cell in indexes[i].tolist():
            masks[i, 0, cell // n_cells_side, cell % n_cells_side] = 0
    return masks


def sample_cells_inverse_cdf(probs, uniforms):
    """The LAW of np.random.choice(K, k, p, replace=False) as sequential inverse-CDF draws: cell j is drawn from p
    with the already drawn cells zeroed and renormalised (P(a then b) = p_a p_b / (1 - p_a), which is what numpy's
    draw-k / keep-the-unique / redraw loop yields).  probs n x K float64, uniforms n x k in (0,1) -> n x k int64.
    The engine's device sampler (pa_sample_dropout_masks) is checked against this with its own uniforms."""
    import numpy as np
    probs = np.asarray(probs, dtype=np.float64)
    n, K = probs.shape
    k = uniforms.shape[1]
    out = np.zeros((n, k), dtype=np.int64)
    for i in range(n):
        p = probs[i].copy()
        for j in range(k):
            cdf = np.cumsum(p)
            pick = int(np.searchsorted(cdf, uniforms[i, j] * cdf[-1], side='right'))
            pick = min(pick, K - 1)
            while p[pick] == 0.0 and pick > 0:      # u*total landed on a zeroed cell's flat step (measure zero)
                pick -= 1
            out[i, j] = pick
            p[pick] = 0.0
    return out


def create_hg(num_stacks, num_modules, num_classes, chan):
    # models/asn_stacked_hg.py:344-347
    return HourglassNet(num_modules=num_modules, num_stacks=num_stacks,
                        chan=chan, num_classes=num_classes)


class ASN(nn.Module):
    """Scale/rotation agent (is_aug) or occlusion agent (is_dropout): the same trunk, a two-Linear head on the
    average-pooled 4x4 map or a 1x1 conv giving one logit per cell (:349-439)."""

    def __init__(self, chan_in, chan_out, scale_num=None, rotation_num=None, is_aug=True, is_dropout=False):
        super().__init__()
        assert is_aug != is_dropout          # :351
        for k in ('skip1', 'skip2', 'skip3', 'skip4', 'neck'):
            setattr(self, 'residual_' + k, Residual(chan_in, chan_out))
        for k in (1, 2, 3, 4):
            setattr(self, 'merge%d' % k, Residual(chan_out, chan_out))
        self.deep_merge = _res_stack(chan_out, 3)
        if is_aug:                  # :373-377
            self.fc_scale = nn.Linear(chan_out, scale_num)
            self.fc_rotation = nn.Linear(chan_out, rotation_num)
        if is_dropout:              # :378-379
            self.out_conv = nn.Conv2d(chan_out, 1, 1, bias=True)
        reference_init_(self)       # Linear layers keep torch's default init (:380-392 skips them)

    def forward(self, feats, is_aug=False, is_dropout=False):
        # :401-439
        assert is_aug != is_dropout
        x = self.residual_skip1(feats['skip1'])
        lower = [self.residual_skip2(feats['skip2']), self.residual_skip3(feats['skip3']),
                 self.residual_skip4(feats['skip4']), self.residual_neck(feats['neck'])]
        for k in (1, 2, 3, 4):
            x = F.max_pool2d(x, 2, 2) + lower[k - 1]
            x = getattr(self, 'merge%d' % k)(x)
        x = self.deep_merge(x)
        if is_dropout:
            return self.out_conv(x)          # n x 1 x 4 x 4 mask logits (:437-439)
        x = F.avg_pool2d(x, 4).flatten(1)
        return self.fc_scale(x), self.fc_rotation(x)


def create_asn(chan_in, chan_out, scale_num=None, rotation_num=None, is_aug=False, is_dropout=False):
    # models/asn_stacked_hg.py:441-444
    return ASN(chan_in, chan_out, scale_num, rotation_num, is_aug=is_aug, is_dropout=is_dropout)


# ---------------------------------------------------------------------------
# deterministic parameter fill shared by oracle tests, goldens and GPU tests.
# Not torch RNG (streams differ across versions): a numpy PCG64 stream per
# tensor, keyed by the tensor's index in state_dict order.
# ---------------------------------------------------------------------------
def deterministic_fill_(module, seed=0):
    import numpy as np
    sd = module.state_dict()
    for idx, (name, t) in enumerate(sd.items()):
        if name.endswith('num_batches_tracked'):
            t.zero_()
            continue
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        shape = tuple(t.shape)
        if name.endswith('running_var'):
            v = rng.uniform(0.5, 1.5, size=shape)
        elif name.endswith('running_mean'):
            v = rng.uniform(-0.1, 0.1, size=shape)
        elif t.dim() == 4 or t.dim() == 2:       # conv / linear weight
            fan = int(np.prod(shape[1:]))
            b = 1.0 / math.sqrt(fan)
            v = rng.uniform(-b, b, size=shape)
        elif '.bn' in name or name.startswith('bn') or _is_bn_key(module, name):
            v = rng.uniform(0.25, 1.0, size=shape) if name.endswith('weight') \
                else rng.uniform(-0.1, 0.1, size=shape)
        else:                                    # conv / linear bias
            v = rng.uniform(-0.05, 0.05, size=shape)
        t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module


def _is_bn_key(module, name):
    parent = name.rsplit('.', 1)[0]
    m = module
    for part in parent.split('.'):
        m = getattr(m, part) if not part.isdigit() else m[int(part)]
    return isinstance(m, nn.BatchNorm2d)
