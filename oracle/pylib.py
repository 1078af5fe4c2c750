"""Oracle (test infrastructure, see oracle/__init__.py): numpy / torch-CPU
restatement of the reference's pose library for the hot path --
Gaussian heatmaps, similarity transforms, argmax + PCKh, losses and the
agent's reward shaping.  Every function cites the reference lines it follows.
Loops are kept scalar and literal on purpose: this is the checker.
"""
import math

import numpy as np
import torch


# ---------------------------------------------------------------- HumanPts
def draw_gaussian(img, pt, sigma=1):
    """pylib/HumanPts.py:82-116.  7x7 patch exp(-(dx^2+dy^2)/tmp_size^2) with
    tmp_size = ceil(3 sigma) = 3 (so the divisor is 9, not 2 sigma^2); the patch's
    top-left is int(pt - 3) -- int() truncates toward zero; assignment, not max."""
    tmp = math.ceil(3 * sigma)
    ulx, uly = int(pt[0] - tmp), int(pt[1] - tmp)
    brx, bry = int(pt[0] + tmp), int(pt[1] + tmp)
    H, W = img.shape
    if ulx >= W or uly >= H or brx < 0 or bry < 0:
        return img
    size = 2 * tmp + 1
    ax = np.arange(0, size, 1, float)
    c = size // 2
    g = np.exp(-((ax[None, :] - c) ** 2 + (ax[:, None] - c) ** 2) / float(tmp ** 2))
    gx0 = max(0, -ulx); gx1 = min(brx + 1, W) - max(0, ulx) + max(0, -ulx)
    gy0 = max(0, -uly); gy1 = min(bry + 1, H) - max(0, uly) + max(0, -uly)
    ix0, ix1 = max(0, ulx), min(brx + 1, W)
    iy0, iy1 = max(0, uly), min(bry + 1, H)
    img[iy0:iy1, ix0:ix1] = g[gy0:gy1, gx0:gx1]
    return img


def pts2heatmap(pts, heatmap_shape, sigma=1):
    """pylib/HumanPts.py:36-46.  pts: n x 2 (x, y) in heat-map pixels."""
    n = pts.shape[0]
    hm = np.zeros((n, heatmap_shape[0], heatmap_shape[1]))
    valid = np.zeros(pts.shape)
    for i in range(n):
        x, y = pts[i][0], pts[i][1]
        if x <= 0 or y <= 0 or x > heatmap_shape[1] or y > heatmap_shape[0]:
            continue
        hm[i] = draw_gaussian(hm[i], pts[i], sigma)
        valid[i] = pts[i]
    return hm, valid


# ---------------------------------------------------------------- HumanAug
def _scalar(v):
    return float(np.asarray(v, dtype=np.float64).reshape(-1)[0])


def get_transform(center, scale, rot, res, size=200):
    """pylib/HumanAug.py:10-35 (identical at pylib/Evaluation.py:213-238)."""
    scale, rot = _scalar(scale), _scalar(rot)
    h = size * scale
    t = np.zeros((3, 3))
    t[0, 0] = float(res) / h
    t[1, 1] = float(res) / h
    t[0, 2] = res * (-float(center[0]) / h + .5)
    t[1, 2] = res * (-float(center[1]) / h + .5)
    t[2, 2] = 1
    if not rot == 0:
        r = -rot * np.pi / 180
        sn, cs = np.sin(r), np.cos(r)
        rm = np.array([[cs, -sn, 0], [sn, cs, 0], [0, 0, 1.0]])
        tm = np.eye(3); tm[0, 2] = -res / 2; tm[1, 2] = -res / 2
        ti = tm.copy(); ti[:2, 2] *= -1
        t = ti.dot(rm.dot(tm.dot(t)))        # same association order as the reference (:34)
    return t


def transform_pts(pts, center, scale, rot, res, size=200, invert=0):
    """pylib/HumanAug.py:45-54 (0-based, float result)."""
    t = get_transform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    p = np.concatenate((pts, np.ones((pts.shape[0], 1))), axis=1).T
    return t.dot(p)[0:2, :].T


def transform_pts_eval(pts, center, scale, rot, res, size=200, invert=0):
    """pylib/Evaluation.py:240-248 (1-based in and out, astype(int) truncation)."""
    t = get_transform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    p = np.concatenate((pts - 1, np.ones((pts.shape[0], 1))), axis=1).T
    return t.dot(p)[0:2, :].T.astype(int) + 1


def shufflelr(pts, width):
    """pylib/HumanAug.py:236-257 (MPII pairs)."""
    pts = pts.copy()
    pts[:, 0] = width - pts[:, 0]
    for a, b in ((0, 5), (1, 4), (2, 3), (10, 15), (11, 14), (12, 13)):
        pts[[a, b]] = pts[[b, a]]
    return pts


_FLIP_PAIRS = ((1, 4), (0, 5), (12, 13), (11, 14), (10, 15), (2, 3))   # pylib/HumanAug.py:182


def flip_heatmaps(maps):
    """pylib/HumanAug.py:198-210 then :179-196: mirror W and swap L/R channels (n x c x h x w)."""
    out = maps.flip(-1).clone()
    for a, b in _FLIP_PAIRS:
        tmp = out[:, a].clone(); out[:, a] = out[:, b]; out[:, b] = tmp
    return out


def warp_bilinear(img, center, scale, rot, res, size=200, flip=False, gain=(1., 1., 1.)):
    """Device-warp specification (the thing the HIP kernel must equal bit-for-tolerance):
    out[c, v, u] = clamp(gain_c * bilinear(img_c, T^-1 (u, v)), 0, 1), zero outside the
    frame, with T = get_transform(center, scale, rot, res) and optional horizontal flip
    of the source frame applied first (data/mpii_for_mpii.py:126-135).  This is the
    pure inverse-affine sampler SURVEY.md section 8(c) specifies in place of the reference's
    scipy.misc/PIL crop (pylib/HumanAug.py:117-176), whose pixels are unpinned.
    img: H x W x 3 float in [0,1] (or uint8 / 255).  Returns 3 x res x res float32."""
    H, W = img.shape[:2]
    src = img.astype(np.float64)
    if src.max() > 1.0 + 1e-9:
        src = src / 255.0
    if flip:
        src = src[:, ::-1]
    tinv = np.linalg.inv(get_transform(center, scale, rot, res, size))
    out = np.zeros((3, res, res), dtype=np.float32)
    for v in range(res):
        for u in range(res):
            sx = tinv[0, 0] * u + tinv[0, 1] * v + tinv[0, 2]
            sy = tinv[1, 0] * u + tinv[1, 1] * v + tinv[1, 2]
            x0 = math.floor(sx); y0 = math.floor(sy)
            fx = sx - x0; fy = sy - y0
            acc = np.zeros(3)
            for dy, wy in ((0, 1 - fy), (1, fy)):
                for dx, wx in ((0, 1 - fx), (1, fx)):
                    xx, yy = x0 + dx, y0 + dy
                    if 0 <= xx < W and 0 <= yy < H:
                        acc += wy * wx * src[yy, xx]
            out[:, v, u] = np.clip(acc * np.asarray(gain), 0, 1)
    return out


# -------------------------------------------------------------- Evaluation
def get_preds(scores):
    """pylib/Evaluation.py:6-23.  1-based (x, y); zero where max <= 0."""
    assert scores.dim() == 4
    n, c, h, w = scores.shape
    maxval, idx = torch.max(scores.reshape(n, c, -1), 2)
    idx = idx.view(n, c, 1) + 1
    preds = idx.repeat(1, 1, 2).float()
    preds[:, :, 0] = (preds[:, :, 0] - 1) % w + 1
    preds[:, :, 1] = torch.floor((preds[:, :, 1] - 1) / h) + 1     # the reference divides by size(2)
    mask = maxval.view(n, c, 1).gt(0).repeat(1, 1, 2).float()
    return preds * mask


def calc_dists(preds, target, normalize, use_zero=False):
    """pylib/Evaluation.py:25-39."""
    preds = preds.float(); target = target.float()
    n, c = preds.shape[0], preds.shape[1]
    dists = torch.zeros(c, n)
    boundary = 0 if use_zero else 1
    for i in range(n):
        for j in range(c):
            if target[i, j, 0] > boundary and target[i, j, 1] > boundary:
                dists[j, i] = torch.dist(preds[i, j, :], target[i, j, :]) / float(normalize[i])
            else:
                dists[j, i] = -1
    return dists


def dist_acc(dists, thr=0.5):
    """pylib/Evaluation.py:41-52."""
    valid = dists.ne(-1)
    if valid.sum() > 0:
        return float(dists.le(thr).eq(valid).sum()) * 1.0 / float(valid.sum())
    return -1


def _acc_from_dists(dists, idxs):
    acc = torch.zeros(len(idxs) + 1)
    s, cnt = 0.0, 0
    for i in range(len(idxs)):
        acc[i + 1] = dist_acc(dists[int(idxs[i])])
        if acc[i + 1] >= 0:
            s = s + acc[i + 1]; cnt += 1
    if cnt != 0:
        acc[0] = s / cnt
    return acc


def accuracy(output, target, idxs, thr=0.5):
    """pylib/Evaluation.py:54-75.  Heat-map-space PCK, normaliser W/10."""
    preds = get_preds(output); gts = get_preds(target)
    norm = torch.ones(preds.size(0)) * output.size(3) / 10
    return _acc_from_dists(calc_dists(preds, gts, norm), idxs)


PCKH_JOINTS = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15]          # pylib/Evaluation.py:81


def final_preds(output, center, scale, res, rot):
    """pylib/Evaluation.py:169-193 (+ transform_preds :195-211)."""
    coords = get_preds(output)
    for n in range(coords.size(0)):
        for p in range(coords.size(1)):
            hm = output[n][p]
            px = int(math.floor(coords[n][p][0])); py = int(math.floor(coords[n][p][1]))
            if 1 < px < res[0] and 1 < py < res[1]:
                diff = torch.tensor([float(hm[py - 1][px] - hm[py - 1][px - 2]),
                                     float(hm[py][px - 1] - hm[py - 2][px - 1])])
                coords[n][p] += diff.sign() * .25
    coords += 0.5
    preds = coords.clone()
    for i in range(coords.size(0)):
        p = transform_pts_eval(coords[i].numpy().astype(np.float64), np.asarray(center[i], dtype=np.float64),
                               scale[i], rot[i], res[0], 200, invert=1)
        preds[i] = torch.from_numpy(p.astype(np.float32))
    return preds


def accuracy_origin_res(output, center, scale, res, grnd_pts, normalizers, rot):
    """pylib/Evaluation.py:77-97."""
    pred_pts = final_preds(output, center, scale, res, rot)
    dists = calc_dists(pred_pts, grnd_pts, normalizers, use_zero=True)
    return _acc_from_dists(dists, PCKH_JOINTS)


def per_person_pckh(output, grnd_heatmap, center, scale, res, grnd_pts, normalizers, rot, thr=0.5):
    """pylib/Evaluation.py:99-167."""
    idxs = torch.tensor(PCKH_JOINTS, dtype=torch.long)
    pred_pts = final_preds(output, center, scale, res, rot)
    n = pred_pts.size(0)
    dists = calc_dists(pred_pts, grnd_pts, normalizers, use_zero=True)
    g = get_preds(grnd_heatmap)
    ind = torch.zeros(pred_pts.size(1), n)
    for i in range(n):
        for c in range(pred_pts.size(1)):
            if g[i, c, 0] > 1 and g[i, c, 1] > 1:
                ind[c, i] = 1
    acc = torch.zeros(n)
    for i in range(n):
        d = dists[:, i].index_select(0, idxs)
        m = ind[:, i].index_select(0, idxs)
        if d.ne(-1).sum() > 0 and m.ne(0).sum() > 0:
            both = d.ne(-1) & m.ne(0)
            ok = d.le(thr) & both
            acc[i] = float(ok.sum()) / float(both.sum())
        else:
            acc[i] = 0
    return acc


# ---------------------------------------------------------------- HumanAcc
def approx_pckh(pred, target, idxs, res):
    """pylib/HumanAcc.py:7-44.  NB py2: normalize = res/10 is INTEGER division (:14)."""
    assert pred.size() == target.size()
    target = target.float(); pred = pred.float()
    dists = torch.zeros((pred.size(1), pred.size(0)))
    normalize = res // 10
    for i in range(pred.size(1)):
        for j in range(pred.size(0)):
            if target[j][i][0] > 0 and target[j][i][1] > 0:
                dists[i][j] = torch.dist(target[j][i], pred[j][i]) / normalize
            else:
                dists[i][j] = -1
    avg, bad = 0.0, 0
    for i in range(len(idxs)):
        a = dist_acc(dists[int(idxs[i])])
        if a >= 0:
            avg += a
        else:
            bad += 1
    return avg / (len(idxs) - bad)


def _acc_dists(pred, target, normalize):
    """The distance matrix every HumanAcc function builds first (e.g. pylib/HumanAcc.py:12-20): [n_joints][n_samples],
    ||target - pred|| / normalize where both target coordinates are > 0, else -1.  normalize: scalar or [n_samples]."""
    assert pred.size() == target.size()
    target = target.float(); pred = pred.float()
    d = (target - pred).pow(2).sum(-1).sqrt().t()                      # [J][B]
    nm = torch.as_tensor(normalize, dtype=torch.float32).reshape(1, -1)
    d = d / nm
    valid = ((target[..., 0] > 0) & (target[..., 1] > 0)).t()
    return torch.where(valid, d, torch.full_like(d, -1.0))


def _joint_acc(row, thr=0.5):
    """per-joint accuracy of one distance row, -1 when the row has no valid entry (pylib/HumanAcc.py:26-37)."""
    valid = row.ne(-1)
    if int(valid.sum()) == 0:
        return -1.0
    return float((row.le(thr) & valid).sum()) / float(valid.sum())


def approx_pckh_per(pred, target, idxs, res):
    """pylib/HumanAcc.py:46-84 -> (average over the joints with a valid sample, per-joint accuracies, -1 = none valid)."""
    dists = _acc_dists(pred, target, res // 10)
    pck = torch.tensor([_joint_acc(dists[int(i)]) for i in idxs], dtype=torch.float32)
    good = pck.ge(0)
    return float(pck[good].sum() / good.sum()), pck


def pckh_report(pred, target, normalizer):
    """pylib/HumanAcc.py:86-137 (PCKh prints, returns nothing): (per-joint PCKh [n], 7 body-part means in the printed
    order Head, Knee, Ankle, Shoulder, Elbow, Wrist, Hip, average over the joints with a valid sample)."""
    dists = _acc_dists(pred, target, torch.as_tensor(normalizer, dtype=torch.float32).reshape(-1))
    pck = torch.tensor([_joint_acc(dists[i]) for i in range(dists.size(0))], dtype=torch.float32)
    parts = [[8, 9], [1, 4], [0, 5], [12, 13], [11, 14], [10, 15], [2, 3]]
    part_means = torch.tensor([(float(pck[a]) + float(pck[b])) / 2 for a, b in parts])
    good = pck.ge(0)
    return pck, part_means, float(pck[good].sum() / good.sum())


def approx_pckh_samples(pred, target, res):
    """pylib/HumanAcc.py:139-177: number of correctly predicted joints per sample."""
    dists = _acc_dists(pred, target, res // 10)
    return (dists.le(0.5) & dists.ne(-1)).sum(0).float()


def correct_predicted_joints(pred, target, res):
    """pylib/HumanAcc.py:179-218: uint8 mask [n_samples][n_joints]."""
    dists = _acc_dists(pred, target, res // 10)
    return (dists.le(0.5) & dists.ne(-1)).t().to(torch.uint8)


def correct_predicted_joints_original_resolution(pred, target, normalizer):
    """pylib/HumanAcc.py:220-259: same with a caller-given scalar normaliser."""
    dists = _acc_dists(pred, target, float(normalizer))
    return (dists.le(0.5) & dists.ne(-1)).t().to(torch.uint8)


def predicted_joints_dist_to_grnd(pred, target, res):
    """pylib/HumanAcc.py:261-308: mean normalised distance over the valid joints of each sample (0 if none)."""
    dists = _acc_dists(pred, target, res // 10)
    valid = dists.ne(-1)
    n = valid.sum(0)
    tot = torch.where(valid, dists, torch.zeros_like(dists)).sum(0)
    return torch.where(n > 0, tot / n.clamp(min=1), torch.zeros_like(tot))


# --------------------------------------------------------------- Criterion
def weighted_l2(pred, gt, weight):
    """pylib/Criterion.py:12-18."""
    loss = (pred - gt) ** 2 * weight
    return loss.sum() / loss.numel()


def stack_mse(outputs, target):
    """stack-hg.py:156-159: sum over stacks of ((o - t)^2).sum()/numel."""
    total = 0
    for o in outputs:
        d = (o - target) ** 2
        total = total + d.sum() / d.numel()
    return total


# -------------------------------------------------------------- utils/util
def gen_groundtruth(pred_distri, indexes, pckh_regular, pckh_agent):
    """utils/util.py:147-253.  pred_distri: n x K probabilities; indexes: n x m sampled bins."""
    g = pred_distri.detach().clone()
    K = g.size(1)
    thres = (1. / K) * 2
    for k in range(len(pckh_regular)):
        others = list(range(K))
        if pckh_regular[k] - pckh_agent[k] > 0:
            inc = 0
            for j in range(indexes.size(1)):
                b = int(indexes[k, j])
                d = 0.2 * g[k, b]
                inc += d
                g[k, b] += d
                others.remove(b)
            for t in others:
                g[k, t] -= inc / len(others)
        else:
            dec = 0
            for j in range(indexes.size(1)):
                b = int(indexes[k, j])
                d = 0.5 * g[k, b]
                dec += d
                g[k, b] -= d
                others.remove(b)
            for t in others:
                g[k, t] += dec / len(others)
        over, under = 0, 0
        not_over = list(range(K)); not_under = list(range(K))
        for t in range(K):
            if g[k, t] > thres:
                over += g[k, t] - thres
                g[k, t] = thres
                not_over.remove(t)
            elif g[k, t] < 0:
                under += g[k, t]
                g[k, t] = 0
                not_under.remove(t)
        gap = over + under
        if gap > 0:
            for t in not_over:
                g[k, t] += gap / len(not_over)
        elif gap < 0:
            for t in not_under:
                g[k, t] += gap / len(not_under)
                if g[k, t] < 0:
                    g[k, t] = 0
        g[k] /= g[k].sum()
    return g


def adjust_lr_value(lr, epoch):
    """utils/util.py:105-117: x0.2 at epoch 100, x0.5 at epoch 140 (applied to opt.lr in place)."""
    if epoch == 100:
        return lr * 0.2
    if epoch == 140:
        return lr * 0.5
    return lr


# ------------------------------------------------ augmentation laws (a10)
SCALE_MEANS = np.arange(-0.6, 0.61, 0.2)        # data/joint_train_s_r_agent.py:33
ROT_MEANS = np.arange(-60, 61, 20)              # data/joint_train_s_r_agent.py:35


def bounded_gaussian(x, z):
    """data/mpii_for_mpii.py:12-13 with the normal draw z passed in."""
    return max(-2 * x, min(2 * x, z * x))


def small_gaussian(mean, var, z):
    """data/joint_train_s_r_agent.py:15-16 with the normal draw z passed in."""
    return max(mean - var + 1e-3, min(mean + var, mean + z * var))


def regular_aug(c, s, width, z_scale, z_rot, u_rot, u_flip, u_gain):
    """data/mpii_for_mpii.py:119-135 with every np.random draw passed in (call order of the reference: randn for the scale,
    randn for the rotation, uniform for "no rotation", random for the flip, three uniforms for the gains).  c, s are the
    torch.FloatTensor quantities of :95-104.  Returns (c', s' (fp32), r, flip, gains)."""
    c = torch.tensor([float(c[0]), float(c[1])], dtype=torch.float32)
    s = torch.tensor([float(s)], dtype=torch.float32)
    s = s * (2 ** bounded_gaussian(0.25, z_scale))
    r = bounded_gaussian(30, z_rot)
    if u_rot <= 0.6:
        r = 0
    flip = u_flip <= 0.5
    if flip:
        c[0] = width - c[0]
    gains = [0.6 + (1.4 - 0.6) * u for u in u_gain]          # np.random.uniform(low, high) = low + (high - low) * random_sample()
    return c.numpy().astype(np.float64), float(s[0]), float(r), bool(flip), gains


def agent_aug(c, s, width, scale_idx, rot_idx, z_scale, z_rot, u_flip, u_gain):
    """data/joint_train_s_r_agent.py:134-139,161-169 (separate_s_r False) with the draws passed in."""
    c = torch.tensor([float(c[0]), float(c[1])], dtype=torch.float32)
    s = torch.tensor([float(s)], dtype=torch.float32)
    f = small_gaussian(SCALE_MEANS[scale_idx], 0.05, z_scale)
    r = small_gaussian(ROT_MEANS[rot_idx], 5, z_rot)
    s = s * (2 ** f)
    flip = u_flip <= 0.5
    if flip:
        c[0] = width - c[0]
    gains = [0.6 + (1.4 - 0.6) * u for u in u_gain]
    return c.numpy().astype(np.float64), float(s[0]), float(r), bool(flip), gains


# ---- one whole dataset sample (the composition of the pieces above)
def _mpii_record(a):
    """data/mpii_for_mpii.py:86-104 (= data/joint_train_s_r_agent.py:105-121): joints, centre, scale, normaliser of one
    annotation entry as the torch.FloatTensor quantities of the reference."""
    pts = torch.tensor(a['joint_self'], dtype=torch.float32)[:, 0:2].clone()
    c = torch.tensor(a['objpos'], dtype=torch.float32)
    s = torch.tensor([a['scale_provided']], dtype=torch.float32)
    assert a['dataset'] == 'MPII'
    c[1] = c[1] + 15 * s[0]
    s = s * 1.25
    return pts, c, s, a['normalizer'] * 0.6


def _img_heatmap(frame_u8, c, s, r, pts, flip, gains, inp_res, out_res, quirk):
    """gen_img_heatmap (data/joint_train_s_r_agent.py:186-205) = data/mpii_for_mpii.py:137-152: crop, joints to heat-map
    coordinates, unannotated joints (x <= 0 or y <= 0 in the image) to 0, Gaussian maps.  c, pts: AFTER the mirror."""
    from . import crop as ocrop
    img = ocrop.source_image(frame_u8, flip, gains)
    inp = ocrop.crop(img, c.numpy(), s.numpy(), r, inp_res, 200, quirk)
    inp = np.ascontiguousarray(inp.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    pts_aug = transform_pts(pts.numpy(), c.numpy(), s.numpy(), r, out_res, 200)
    gone = ((pts[:, 0] <= 0) | (pts[:, 1] <= 0)).numpy()
    pts_aug[gone, :] = 0
    heat, _ = pts2heatmap(pts_aug, [out_res, out_res], sigma=1)
    return inp, heat.astype(np.float32)


def mpii_getitem(frame_u8, a, draws=None, is_train=True, inp_res=256, out_res=64, quirk=True):
    """MPII.__getitem__ (data/mpii_for_mpii.py:83-163) for one annotation entry `a` and its decoded frame, with the
    np.random draws of the call passed in (draws[7]: randn, randn, then five random_sample() -- tests/inputs.legacy_draws).
    Returns (inp [3][res][res] fp32, heatmap [16][64][64] fp32, c [2], s [1], r [1], pts [16][2], normalizer)."""
    pts, c, s, normalizer = _mpii_record(a)
    r, flip, gains = 0, False, (1., 1., 1.)
    width = frame_u8.shape[1]
    if is_train:
        c_np, s_f, r, flip, gains = regular_aug(c.numpy(), float(s[0]), width, draws[0], draws[1], draws[2], draws[3], draws[4:7])
        c, s = torch.from_numpy(c_np.astype(np.float32)), torch.tensor([s_f], dtype=torch.float32)
        if flip:
            pts = torch.from_numpy(shufflelr(pts.numpy(), width)).float()
    inp, heat = _img_heatmap(frame_u8, c, s, r, pts, flip, gains, inp_res, out_res, quirk)
    return inp, heat, c.numpy(), s.numpy(), np.array([r], dtype=np.float32), pts.numpy(), normalizer


def agent_getitem(frame_u8, a, scale_idx, rot_idx, draws, separate_s_r=False, inp_res=256, out_res=64, quirk=True):
    """AGENT.__getitem__ with the bins given (data/joint_train_s_r_agent.py:98-177).  separate_s_r False: ONE crop at
    (agent scale, agent rotation) behind flip + colour -> the 7-tuple; True: [scale-only crop, rotation-only crop] (no
    flip, no colour; draws[0:2] only) -> two 7-tuples."""
    pts, c, s, normalizer = _mpii_record(a)
    width = frame_u8.shape[1]
    if not separate_s_r:
        c_np, s_f, r, flip, gains = agent_aug(c.numpy(), float(s[0]), width, scale_idx, rot_idx, draws[0], draws[1], draws[2], draws[3:6])
        c2, s2 = torch.from_numpy(c_np.astype(np.float32)), torch.tensor([s_f], dtype=torch.float32)
        if flip:
            pts = torch.from_numpy(shufflelr(pts.numpy(), width)).float()
        inp, heat = _img_heatmap(frame_u8, c2, s2, r, pts, flip, gains, inp_res, out_res, quirk)
        return inp, heat, c2.numpy(), s2.numpy(), np.array([r], dtype=np.float32), pts.numpy(), normalizer
    f = small_gaussian(SCALE_MEANS[scale_idx], 0.05, draws[0])
    r_aug = small_gaussian(ROT_MEANS[rot_idx], 5, draws[1])
    s_aug = s * (2 ** f)
    out = []
    for ss, rr in ((s_aug, 0), (s, r_aug)):
        inp, heat = _img_heatmap(frame_u8, c, ss, rr, pts, False, (1., 1., 1.), inp_res, out_res, quirk)
        out.append((inp, heat, c.numpy(), ss.numpy(), np.array([rr], dtype=np.float32), pts.numpy(), normalizer))
    return out
