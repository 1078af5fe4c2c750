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


# ------------------------------------------------------------------ frames and cases for the crop / warp parity (row a9)
def _mark_squares(img, x0, y0):
    """a 28x28 pure-black and a 28x28 pure-white square side by side: a crop that contains both makes scipy's per-image
    min/max byte scaling the identity, even after the >=2x pre-downscale"""
    img[y0:y0 + 28, x0:x0 + 28] = 0
    img[y0:y0 + 28, x0 + 52:x0 + 80] = 255


def warp_frame(kind, H=720, W=1280):
    """uint8 H x W x 3 'MPII-shape' frame: 'smooth' (low-frequency sinusoids), 'noise' (the bench's U{0..255} pixels),
    'checker' (period-3 checkerboard with a diagonal ramp: the aliasing worst case for a point sampler)."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    if kind == 'smooth':
        ch = [127.5 + 100 * np.sin(xx / (37.0 + 11 * k) + 0.3 * k) * np.cos(yy / (29.0 + 7 * k)) + 20 * np.sin((xx + yy) / 9.0 + k)
              for k in range(3)]
        img = np.clip(np.rint(np.stack(ch, -1)), 0, 255).astype(np.uint8)
    elif kind == 'noise':
        img = rng(77, 5).integers(0, 256, (H, W, 3), dtype=np.uint8)
    elif kind == 'checker':
        base = (((xx // 3).astype(np.int64) + (yy // 3).astype(np.int64)) % 2) * 200.0
        ch = [np.clip(base + 0.04 * (k + 1) * (xx + yy) % 56, 0, 255) for k in range(3)]
        img = np.rint(np.stack(ch, -1)).astype(np.uint8)
    else:
        raise ValueError(kind)
    _mark_squares(img, 600, 346)
    _mark_squares(img, 70, 76)
    return img


# (frame kind, centre BEFORE the mirror, scale (final, fp32), rotation deg, flip, colour gains, neutral byte scaling?)
WARP_CASES = [
    ('smooth',  (655.3, 371.8), 2.013, 0.0, 0, (1.0, 1.0, 1.0), True),     # resize 402 -> 256 only
    ('smooth',  (655.3, 371.8), 2.013, 25.0, 0, (1.0, 1.0, 1.0), True),    # + rotation
    ('smooth',  (629.5, 350.2), 3.47, 0.0, 0, (1.0, 1.0, 1.0), True),      # scale_factor 2.71: pre-downscale
    ('smooth',  (640.0, 360.0), 1.2, -40.0, 0, (1.0, 1.0, 1.0), True),     # 240 -> 256 upscale + rotation
    ('noise',   (641.7, 362.4), 2.3, 0.0, 0, (1.0, 1.0, 1.0), True),
    ('noise',   (648.2, 355.9), 2.9, -17.5, 0, (1.0, 1.0, 1.0), True),     # pre-downscale + rotation
    ('noise',   (633.1, 366.6), 5.7, 33.0, 0, (1.0, 1.0, 1.0), True),      # scale_factor 4.45
    ('checker', (640.0, 360.0), 0.9, 0.0, 0, (1.0, 1.0, 1.0), True),       # 180 -> 256 upscale
    ('checker', (644.4, 357.3), 2.5625, 0.0, 0, (1.0, 1.0, 1.0), True),    # scale_factor 2.002: just above the threshold
    ('checker', (644.4, 357.3), 2.55, 8.0, 0, (1.0, 1.0, 1.0), True),      # scale_factor 1.992: just below (widest resize filter)
    ('noise',   (632.0, 352.0), 2.55, 0.0, 1, (0.7, 1.0, 1.35), True),     # mirror + colour gain
    ('smooth',  (651.0, 349.0), 3.1, 58.0, 1, (1.4, 0.6, 1.1), True),
    ('checker', (110.5, 90.25), 2.2, 12.0, 0, (1.0, 1.0, 1.0), True),      # window reaches outside the frame (zero padding)
    ('smooth',  (300.0, 500.0), 2.0, 0.0, 0, (1.0, 1.0, 1.0), False),      # no pure black / white in the crop: the byte-scaling quirk
    ('smooth',  (300.0, 500.0), 3.3, -20.0, 0, (1.0, 1.0, 1.0), False),    # the quirk after the pre-downscale
]


# ---- a 3-person MPII-format set (JSON + image files) for the whole-sample golden (tests/golden/dataset.npz)
DATASET_FRAMES = (('smooth', 540, 960), ('checker', 480, 640), ('smooth', 600, 800))


def dataset_people():
    """(frames uint8, annotation list in the JSON format data/mpii_for_mpii.py:28-42 reads).  Person 0, 1: training split,
    person 2: validation.  Joints around the person's centre at the person's scale; one joint of person 0 is unannotated
    (0, 0) and one of person 1 lies left of the frame (x < 0): both take the `pts <= 0` branch of :144-147."""
    frames = [warp_frame(kind, H, W) for kind, H, W in DATASET_FRAMES]
    g = rng(4711)
    anno = []
    for i, (f, (cx, cy, sc)) in enumerate(zip(frames, ((470.0, 250.0, 1.9), (300.5, 228.25, 1.35), (410.0, 310.0, 3.1)))):
        joints = np.stack([cx + g.normal(0, 1, 16) * 40 * sc, cy + g.normal(0, 1, 16) * 55 * sc, np.ones(16)], 1)
        joints = np.round(joints, 2)
        _mark_squares(f, int(cx) - 40, int(cy) + 10)          # pure black + pure white inside every crop window of this person
        if i == 0:
            joints[6] = 0.0
        if i == 1:
            joints[3, 0] = -12.5
        anno.append({'dataset': 'MPII', 'isValidation': 1.0 if i == 2 else 0.0, 'img_paths': 'person%d.png' % i,
                     'objpos': [cx, cy], 'scale_provided': sc, 'joint_self': joints.tolist(), 'normalizer': 60.0 * sc + 0.5 * i})
    return frames, anno


def write_dataset(folder):
    """Write the set as <folder>/mpii-hr-lsp-normalizer.json + PNG files (lossless: every decoder returns the same bytes)."""
    import json
    import os
    from PIL import Image
    frames, anno = dataset_people()
    os.makedirs(folder, exist_ok=True)
    for a, f in zip(anno, frames):
        Image.fromarray(f).save(os.path.join(folder, a['img_paths']))
    path = os.path.join(folder, 'mpii-hr-lsp-normalizer.json')
    with open(path, 'w') as fd:
        json.dump(anno, fd)
    return path, frames, anno


def legacy_draws(seed):
    """The np.random draws of ONE __getitem__ in the reference's call order, as the raw numbers pa_sample_aug_given takes:
    randn (scale), randn (rotation), random_sample x 5 (rotation forced to 0, flip, three gains) -- RandomState.uniform(lo, hi) is
    lo + (hi - lo) * random_sample(), so a second RandomState with the same seed yields the same stream."""
    st = np.random.RandomState(seed)
    return np.array([st.randn(), st.randn()] + [st.random_sample() for _ in range(5)], dtype=np.float64)
