"""pylib/HumanPts.py of the reference, on the GPU: Gaussian heat-map targets."""
from ._dev import lib, check, ptr, stream, dev, to_dev, np, torch


def pts2heatmap_batch(pts, height, width):
    """pts: [B][J][2] (x, y) heat-map coordinates, float64 -> GPU tensor [B][J][H][W] fp32.
    Reference: pts2heatmap (pylib/HumanPts.py:36-46) + draw_gaussian (:82-116), batched."""
    p = to_dev(pts, torch.float64)
    B, J = p.shape[0], p.shape[1]
    out = torch.empty((B, J, height, width), dtype=torch.float32, device=dev())
    check(lib().pa_gaussian_heatmap(ptr(p), ptr(out), B, J, height, width, stream()), 'pa_gaussian_heatmap')
    return out


def pts2heatmap(pts, heatmap_shape, sigma=1):
    """Same signature and return values as the reference (pylib/HumanPts.py:36): pts n x 2 ->
    (heatmap n x H x W, valid_pts n x 2), both numpy."""
    if sigma != 1:
        raise ValueError('only sigma == 1 (7x7 patch) is used by the reference training path')
    pts = np.asarray(pts, dtype=np.float64)
    hm = pts2heatmap_batch(pts[None], int(heatmap_shape[0]), int(heatmap_shape[1]))[0]
    skip = (pts[:, 0] <= 0) | (pts[:, 1] <= 0) | (pts[:, 0] > heatmap_shape[1]) | (pts[:, 1] > heatmap_shape[0])
    valid = np.where(skip[:, None], 0.0, pts)
    return hm.double().cpu().numpy(), valid


def draw_gaussian(img, pt, sigma=1):
    """pylib/HumanPts.py:82: returns the map with the 7x7 Gaussian pasted (numpy in, numpy out).
    Unlike pts2heatmap this does not apply the validity rule of :41-43."""
    H, W = img.shape
    p = np.asarray(pt, dtype=np.float64).reshape(1, 1, 2).copy()
    # draw_gaussian itself has no x<=0 test: shift the validity rule out of the way by evaluating on a
    # canvas translated by +8 pixels, then cropping
    big = pts2heatmap_batch(p + 8.0, H + 16, W + 16)[0, 0, 8:8 + H, 8:8 + W]
    out = np.array(img, dtype=np.float64, copy=True)
    patch = big.double().cpu().numpy()
    out[patch > 0] = patch[patch > 0]
    return out
