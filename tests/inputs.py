"""Seeded synthetic inputs shared by the golden-vector generator, the oracle
tests and the GPU parity tests.  numpy PCG64 only (stable across versions)."""
import numpy as np


def rng(*key):
    return np.random.Generator(np.random.PCG64(list(key)))


def images(seed, n, res):
    """n x 3 x res x res float32 in [0,1)."""
    return rng(seed, 1).random((n, 3, res, res), dtype=np.float32)


def heat_pts(seed, n, joints=16, res=64, invalid_frac=0.1):
    """n x joints x 2 (x, y) float64 heat-map coordinates, some set to (0,0) = invalid,
    a few pushed onto the borders to exercise clipping."""
    g = rng(seed, 2)
    pts = g.uniform(4.0, res - 4.0, size=(n, joints, 2))
    edge = g.random((n, joints)) < 0.15
    pts[edge] = g.uniform(0.2, 3.5, size=(int(edge.sum()), 2))
    far = g.random((n, joints)) < 0.1
    pts[far] = g.uniform(res - 3.5, res + 0.9, size=(int(far.sum()), 2))
    bad = g.random((n, joints)) < invalid_frac
    pts[bad] = 0.0
    return pts


def heatmaps_from_pts(pts, res=64):
    from oracle import pylib
    import numpy as np
    out = np.zeros((pts.shape[0], pts.shape[1], res, res), dtype=np.float32)
    for i in range(pts.shape[0]):
        hm, _ = pylib.pts2heatmap(pts[i], [res, res], sigma=1)
        out[i] = hm.astype(np.float32)
    return out


def noisy_heatmaps(seed, target, noise=0.15):
    """'Network output'-like maps: target + noise, so argmax lands near but not on the GT."""
    g = rng(seed, 3)
    return (target + noise * g.standard_normal(target.shape)).astype(np.float32)


def person_meta(seed, n, joints=16):
    """MPII-shape annotations: centre, scale (already x1.25), rotation, joints in image px, normaliser."""
    g = rng(seed, 4)
    c = np.stack([g.uniform(400, 880, n), g.uniform(250, 470, n)], 1)
    s = g.uniform(1.5, 3.5, n) * 1.25
    r = np.where(g.random(n) < 0.5, 0.0, g.uniform(-60, 60, n))
    pts = c[:, None, :] + g.normal(0, 1, (n, joints, 2)) * (60 * s)[:, None, None]
    pts[..., 0] = np.clip(pts[..., 0], 1.0, 1279.0)      # joints clipped to the 1280x720 frame
    pts[..., 1] = np.clip(pts[..., 1], 1.0, 719.0)
    bad = g.random((n, joints)) < 0.1
    pts[bad] = 0.0
    norm = g.uniform(40, 120, n) * 0.6
    return c, s, r, pts, norm
