"""pylib/HumanAcc.py of the reference, on the GPU.

Every function of that file first builds the matrix  dists[joint][sample] = ||target - pred|| / normalize  where both
target coordinates are > 0 and -1 elsewhere (e.g. :12-20), with `normalize = res/10` under Python-2 INTEGER division
(64 -> 6, not 6.4).  Here the matrix comes from the pa_pck kernel (include/poseadv.h) and the few reductions on top of
it are device tensor ops; nothing is copied to the host except the scalars the reference returns as Python floats."""
from ._dev import lib, check, ptr, stream, dev, to_dev, torch


def _dists(pred, target, normalize):
    """[n_joints][n_samples] on the device; normalize: Python scalar or per-sample tensor."""
    p = to_dev(pred, torch.float32)
    t = to_dev(target, torch.float32)
    assert p.shape == t.shape
    B, J = p.shape[0], p.shape[1]
    if isinstance(normalize, (int, float)):
        norm = torch.full((B,), float(normalize), dtype=torch.float32, device=dev())
    else:
        norm = to_dev(normalize, torch.float32).reshape(-1)
    ix = torch.zeros(1, dtype=torch.int32, device=dev())
    acc = torch.zeros(2, dtype=torch.float32, device=dev())
    d = torch.empty((J, B), dtype=torch.float32, device=dev())
    check(lib().pa_pck(ptr(p), ptr(t), ptr(norm), 0.0, ptr(ix), 1, 0.5, None, B, J, ptr(acc), None, ptr(d), stream()), 'pa_pck')
    return d


def approx_PCKh(pred, target, idxs, res):
    """pylib/HumanAcc.py:7-44.  pred/target: b x n x 2; valid where target > 0; normaliser res/10 with
    the reference's Python-2 INTEGER division (64 -> 6); threshold 0.5; mean over joints with >=1 valid."""
    p = to_dev(pred, torch.float32)
    t = to_dev(target, torch.float32)
    assert p.shape == t.shape
    B, J = p.shape[0], p.shape[1]
    norm = torch.full((B,), float(int(res) // 10), dtype=torch.float32, device=dev())
    ix = torch.as_tensor(list(idxs), dtype=torch.int32, device=dev())
    acc = torch.zeros(len(idxs) + 1, dtype=torch.float32, device=dev())
    check(lib().pa_pck(ptr(p), ptr(t), ptr(norm), 0.0, ptr(ix), len(idxs), 0.5, None, B, J, ptr(acc), None, None, stream()), 'pa_pck')
    return float(acc[0])


def _joint_acc(d, thr=0.5):
    """rows of d -> accuracy per row, -1 where a row has no valid entry."""
    valid = d.ne(-1)
    n = valid.sum(1)
    ok = (d.le(thr) & valid).sum(1)
    return torch.where(n > 0, ok.float() / n.clamp(min=1).float(), torch.full_like(ok, -1, dtype=torch.float32))


def approx_PCKh_per(pred, target, idxs, res):
    """pylib/HumanAcc.py:46-84 -> (avg_acc, pckhs[len(idxs)]; -1 for a joint without a valid sample)."""
    d = _dists(pred, target, int(res) // 10)
    pck = _joint_acc(d[torch.as_tensor([int(i) for i in idxs], device=d.device)])
    good = pck.ge(0)
    return float(pck[good].sum() / good.sum()), pck


PART_NAMES = ('Head', 'Knee', 'Ankle', 'Shoulder', 'Elbow', 'Wrist', 'Hip')                  # pylib/HumanAcc.py:123
PART_IDXS = ((8, 9), (1, 4), (0, 5), (12, 13), (11, 14), (10, 15), (2, 3))                   # :124


def PCKh(pred, target, normalizer):
    """pylib/HumanAcc.py:86-137: prints the 7 body-part PCKh values and the average (like the reference it returns
    nothing); `normalizer` is per sample."""
    pck = _joint_acc(_dists(pred, target, normalizer)).cpu()
    for name, (a, b) in zip(PART_NAMES, PART_IDXS):
        print('%s: %.4f' % (name, (float(pck[a]) + float(pck[b])) / 2))
    good = pck.ge(0)
    print('Average PCKh is: %.4f' % float(pck[good].sum() / good.sum()))


def approx_PCKh_samples(pred, target, res):
    """pylib/HumanAcc.py:139-177: correctly predicted joints per sample (float counts)."""
    d = _dists(pred, target, int(res) // 10)
    return (d.le(0.5) & d.ne(-1)).sum(0).float()


def correct_predicted_joints(pred, target, res):
    """pylib/HumanAcc.py:179-218: uint8 mask [n_samples][n_joints]."""
    d = _dists(pred, target, int(res) // 10)
    return (d.le(0.5) & d.ne(-1)).t().contiguous().to(torch.uint8)


def correct_predicted_joints_original_resolution(pred, target, normalizer):
    """pylib/HumanAcc.py:220-259: the same with a caller-given scalar normaliser."""
    d = _dists(pred, target, float(normalizer))
    return (d.le(0.5) & d.ne(-1)).t().contiguous().to(torch.uint8)


def predicted_joints_dist_to_grnd(pred, target, res):
    """pylib/HumanAcc.py:261-308: mean normalised distance over the valid joints of each sample (0 if none)."""
    d = _dists(pred, target, int(res) // 10)
    valid = d.ne(-1)
    n = valid.sum(0)
    tot = torch.where(valid, d, torch.zeros_like(d)).sum(0)
    return torch.where(n > 0, tot / n.clamp(min=1), torch.zeros_like(tot))
