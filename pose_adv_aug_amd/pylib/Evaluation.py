"""pylib/Evaluation.py of the reference, on the GPU: arg-max, PCKh, back-projection."""
from ._dev import lib, check, ptr, stream, dev, to_dev, np, torch
from . import HumanAug

PCKH_JOINTS = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15]      # pylib/Evaluation.py:81


def get_preds(scores):
    """pylib/Evaluation.py:6-23: b x n x h x w -> b x n x 2 (1-based x, y; 0 where max <= 0)."""
    s = to_dev(scores, torch.float32)
    assert s.dim() == 4, 'Score maps should be 4-dim'
    B, J, H, W = s.shape
    out = torch.empty((B, J, 2), dtype=torch.float32, device=dev())
    check(lib().pa_get_preds(ptr(s), B, J, H, W, ptr(out), None, stream()), 'pa_get_preds')
    return out


def _pck(pred, gt, norm, boundary, idxs, thr=0.5, vis=None, want_person=False, want_dists=False):
    p = to_dev(pred, torch.float32); g = to_dev(gt, torch.float32); nm = to_dev(norm, torch.float32).reshape(-1)
    B, J = p.shape[0], p.shape[1]
    ix = torch.as_tensor([int(i) for i in idxs], dtype=torch.int32, device=dev())
    acc = torch.zeros(len(idxs) + 1, dtype=torch.float32, device=dev())
    person = torch.zeros(B, dtype=torch.float32, device=dev()) if want_person else None
    dists = torch.zeros((J, B), dtype=torch.float32, device=dev()) if want_dists else None
    v = to_dev(vis, torch.float32) if vis is not None else None
    check(lib().pa_pck(ptr(p), ptr(g), ptr(nm), float(boundary), ptr(ix), len(idxs), float(thr), ptr(v), B, J,
                       ptr(acc), ptr(person), ptr(dists), stream()), 'pa_pck')
    return acc, person, dists


def calc_dists(preds, target, normalize, use_zero=False):
    """pylib/Evaluation.py:25-39 -> n_joints x n_samples matrix, -1 where the target is invalid."""
    J = preds.shape[1]
    _, _, d = _pck(preds, target, normalize, 0 if use_zero else 1, list(range(min(J, 64))), want_dists=True)
    return d


def dist_acc(dists, thr=0.5):
    """pylib/Evaluation.py:41-52 (tiny reduction on whatever device `dists` lives on)."""
    valid = dists.ne(-1)
    if valid.sum() > 0:
        return float((dists.le(thr) & valid).sum()) / float(valid.sum())
    return -1


def accuracy(output, target, idxs, thr=0.5):
    """pylib/Evaluation.py:54-75: PCK in heat-map space, normaliser W/10."""
    o = to_dev(output, torch.float32)
    preds = get_preds(o); gts = get_preds(target)
    norm = torch.ones(preds.size(0), device=dev()) * o.size(3) / 10
    acc, _, _ = _pck(preds, gts, norm, 1, idxs, thr)
    return acc


def final_preds(output, center, scale, res, rot):
    """pylib/Evaluation.py:169-193."""
    o = to_dev(output, torch.float32)
    B, J, H, W = o.shape
    assert H == res[1] and W == res[0]
    c = to_dev(center, torch.float32).reshape(B, 2); s = to_dev(scale, torch.float32).reshape(B)
    r = to_dev(rot, torch.float32).reshape(B)
    out = torch.empty((B, J, 2), dtype=torch.float32, device=dev())
    tmp = torch.empty((B, J, 2), dtype=torch.float32, device=dev())
    check(lib().pa_final_preds(ptr(o), ptr(c), ptr(s), ptr(r), B, J, H, W, ptr(out), ptr(tmp), stream()), 'pa_final_preds')
    return out


def accuracy_origin_res(output, center, scale, res, grnd_pts, normalizers, rot):
    """pylib/Evaluation.py:77-97: head-normalised PCKh at original resolution."""
    pred = final_preds(output, center, scale, res, rot)
    acc, _, _ = _pck(pred, grnd_pts, normalizers, 0, PCKH_JOINTS)
    return acc


def per_person_pckh(output, grnd_heatmap, center, scale, res, grnd_pts, normalizers, rot, thr=0.5):
    """pylib/Evaluation.py:99-167: per-sample PCKh over joints that are annotated AND visible in the crop."""
    pred = final_preds(output, center, scale, res, rot)
    vis = get_preds(grnd_heatmap)
    _, person, _ = _pck(pred, grnd_pts, normalizers, 0, PCKH_JOINTS, thr, vis=vis, want_person=True)
    return person


def transform_preds(coords, center, scale, res, rot):
    """pylib/Evaluation.py:195-211."""
    return torch.from_numpy(TransformPts(np.asarray(coords, dtype=np.float64), np.asarray(center, dtype=np.float64),
                                         scale, rot, res[0], size=200, invert=1))


GetTransform = HumanAug.GetTransform                     # pylib/Evaluation.py:213-238 is a copy


def TransformPts(pts, center, scale, rot, res, size, invert=0):
    """pylib/Evaluation.py:240-248: the 1-based, integer-truncating variant."""
    t = GetTransform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    p = np.concatenate((np.asarray(pts) - 1, np.ones((len(pts), 1))), axis=1).T
    return np.dot(t, p)[0:2, :].T.astype(int) + 1
