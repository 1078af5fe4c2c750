"""Oracle (test infrastructure, see oracle/__init__.py): HumanAug.crop restated in numpy down to the pixel
arithmetic of the libraries beneath it.

The reference's crop (pylib/HumanAug.py:117-176) delegates its pixels to scipy.misc.{imresize, imrotate, toimage}
(scipy < 1.3, `scipy/misc/pilutil.py`; the reference's README pins no version) which are thin wrappers over PIL.
Both layers are restated here:

  * `bytescale` / `toimage` / `imresize` / `imrotate`   -- scipy 0.19..1.2 pilutil.py (absent from this image's scipy 1.15)
  * `pil_resize_bilinear`  -- Pillow `Image.resize(size, BILINEAR)`: src/libImaging/Resample.c (precompute_coeffs,
    normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc: 22-bit fixed point, horizontal pass first)
  * `pil_rotate_bilinear`  -- Pillow `Image.rotate(angle, BILINEAR)`: src/PIL/Image.py rotate() + src/libImaging/Geometry.c
    (affine_transform at pixel centres, bilinear_filter32RGB, zero fill)

Pinned: tests/golden/crop.npz holds outputs of the reference's own crop() run (transliterated, tests/golden/make_goldens.py)
over the real Pillow 12.2 of the build container; tests/test_oracle_golden.py requires `crop` below to reproduce them
BIT-EXACTLY, and -- where Pillow is importable -- the two PIL restatements to equal Pillow on random images.

`crop(..., quirk=False)` is the device specification: scipy's toimage() stretches every float image it is handed to
[min, max] -> [0, 255] (an accident of the host pipeline that partly undoes the colour jitter);
the device warp does not do that (it always scales by the full range), and quirk=False states exactly that.
For crops that contain a zero and a full-range pixel the two are identical.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c


# ------------------------------------------------------------------------------------------------ scipy.misc.pilutil
def bytescale(data, cmin=None, cmax=None, high=255, low=0):
    """pilutil.bytescale: uint8 passes through, everything else is mapped [cmin, cmax] -> [low, high], +0.5, truncated."""
    if data.dtype == np.uint8:
        return data
    if cmin is None:
        cmin = data.min()
    if cmax is None:
        cmax = data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    bytedata = (data - cmin) * scale + low
    return (bytedata.clip(low, high) + 0.5).astype(np.uint8)


def toimage(arr, quirk=True, full=None):
    """pilutil.toimage for an H x W x 3 array -> uint8 H x W x 3 (the bytes PIL would hold).
    quirk=False: scale by the full range `full` (1.0 for [0,1] floats, 255 for byte-valued floats) instead of [min, max]."""
    if quirk:
        return bytescale(arr)
    return bytescale(arr, cmin=0.0, cmax=full)


# ------------------------------------------------------------------------------------------------ Pillow: resize
def _resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc for the bilinear (triangle) filter."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    bounds, coeffs = [], []
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)            # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            wv = 1.0 - v if v < 1.0 else 0.0
            w.append(wv)
            ww += wv
        k = [(wv / ww if ww != 0.0 else wv) for wv in w]
        coeffs.append(np.array([int(0.5 + kv * (1 << PRECISION_BITS)) for kv in k], dtype=np.int64))
        bounds.append((xmin, xmax))
    return bounds, coeffs


def _clip8(ss):
    return np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis1(img, out_size):
    """one pass along axis 1 of an [R][in][3] uint8 image: 22-bit fixed point, rounding constant 2^21, clip8"""
    bounds, coeffs = _resample_coeffs(img.shape[1], out_size)
    out = np.empty((img.shape[0], out_size, img.shape[2]), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx, ((xmin, xmax), k) in enumerate(zip(bounds, coeffs)):
        acc = (src[:, xmin:xmin + xmax, :] * k[None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        out[:, xx, :] = _clip8(acc)
    return out, bounds


def pil_resize_bilinear(img, out_w, out_h):
    """Image.resize((out_w, out_h), BILINEAR) on an H x W x 3 uint8 array (Image.py resize + Resample.c ImagingResample):
    identical size -> copy; horizontal pass first (only over the source rows the vertical pass will read), then vertical."""
    h, w = img.shape[:2]
    if (w, h) == (out_w, out_h):
        return img.copy()
    need_h, need_v = out_w != w, out_h != h
    cur = img
    if need_h:
        if need_v:
            vb, _ = _resample_coeffs(h, out_h)
            first = vb[0][0]
            last = vb[-1][0] + vb[-1][1]
            cur = cur[first:last]
        cur, _ = _resample_axis1(cur, out_w)
        if need_v:                                    # rows outside [first, last) are never read: same result as Pillow's shifted bounds
            full = np.zeros((h, out_w, 3), dtype=np.uint8)
            full[first:last] = cur
            cur = full
    if need_v:
        t, _ = _resample_axis1(np.ascontiguousarray(cur.transpose(1, 0, 2)), out_h)
        cur = np.ascontiguousarray(t.transpose(1, 0, 2))
    return cur


# ------------------------------------------------------------------------------------------------ Pillow: rotate
def pil_rotate_bilinear(img, angle):
    """Image.rotate(angle, resample=BILINEAR) (expand=0, centre = (w/2, h/2), zero fill) on an H x W x 3 uint8 array:
    Image.py rotate() builds the inverse affine matrix (coefficients rounded to 15 decimals), Geometry.c
    ImagingGenericTransform evaluates it at pixel CENTRES and samples with bilinear_filter32RGB."""
    h, w = img.shape[:2]
    angle = angle % 360.0
    if angle == 0:
        return img.copy()
    if angle == 180:
        return np.ascontiguousarray(img[::-1, ::-1])
    if angle in (90, 270) and w == h:
        return np.ascontiguousarray(np.rot90(img, 1 if angle == 90 else 3))
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64) + 0.5, np.arange(w, dtype=np.float64) + 0.5, indexing='ij')
    xin = m[0] * xs + m[1] * ys + m[2]
    yin = m[3] * xs + m[4] * ys + m[5]
    inside = ~((xin < 0.0) | (xin >= w) | (yin < 0.0) | (yin >= h))
    xin = xin - 0.5
    yin = yin - 0.5
    x = np.floor(xin).astype(np.int64)
    y = np.floor(yin).astype(np.int64)
    dx = xin - x
    dy = yin - y
    x0 = np.clip(x, 0, w - 1)
    x1 = np.clip(x + 1, 0, w - 1)
    y0 = np.clip(y, 0, h - 1)
    src = img.astype(np.float64)
    out = np.zeros_like(img)
    has_y1 = (y + 1 >= 0) & (y + 1 < h)
    y1 = np.clip(y + 1, 0, h - 1)
    for b in range(img.shape[2]):
        ch = src[:, :, b]
        v1 = ch[y0, x0] + (ch[y0, x1] - ch[y0, x0]) * dx
        v2 = ch[y1, x0] + (ch[y1, x1] - ch[y1, x0]) * dx
        v2 = np.where(has_y1, v2, v1)
        v = v1 + (v2 - v1) * dy
        out[:, :, b] = np.where(inside, v.astype(np.uint8), 0)            # (UINT8)v1: truncation
    return out


# ------------------------------------------------------------------------------------------------ pilutil wrappers
def imresize_fraction(arr, frac, quirk=True, full=1.0):
    """pilutil.imresize(arr, size=<float>, interp='bilinear'): new size = (im.size * frac).astype(int)."""
    im = toimage(arr, quirk, full)
    h, w = im.shape[:2]
    nw, nh = int(w * frac), int(h * frac)
    return pil_resize_bilinear(im, nw, nh)


def imresize_shape(arr, shape_hw, quirk=True, full=1.0):
    """pilutil.imresize(arr, (h, w)) (default interp='bilinear')."""
    im = toimage(arr, quirk, full)
    return pil_resize_bilinear(im, shape_hw[1], shape_hw[0])


def imrotate(arr, angle, quirk=True, full=1.0):
    """pilutil.imrotate(arr, angle, interp='bilinear')."""
    return pil_rotate_bilinear(toimage(arr, quirk, full), float(angle))


# ------------------------------------------------------------------------------------------------ HumanAug.crop
def _transform_single_pt_inv(pt, center, scale, res, size):
    """pylib/HumanAug.py:37-43 with invert=1 and rot=0: int-TRUNCATED inverse-transformed point."""
    from .pylib import get_transform
    t = np.linalg.inv(get_transform(center, scale, 0, res, size))
    p = np.dot(t, np.array([pt[0], pt[1], 1.]).T)
    return p[:2].astype(int)


def crop_window(center, scale, rot, res, size=200):
    """The integer geometry of pylib/HumanAug.py:117-148 for fp32 centre / scale (torch tensors in the reference):
    returns (scale_factor, center', scale', ul, br, pad) with ul / br already padded for the rotation."""
    center = np.asarray(center, dtype=np.float32).reshape(2)
    scale = np.asarray(scale, dtype=np.float32).reshape(1)
    scale_factor = float(scale[0] * np.float32(size)) / float(res)
    if scale_factor < 2:
        scale_factor = 1
    center = (center / scale_factor).astype(np.float32)
    scale = (scale / scale_factor).astype(np.float32)
    ul = _transform_single_pt_inv([0, 0], center, scale, res, size)
    br = _transform_single_pt_inv([res, res], center, scale, res, size)
    if scale_factor >= 2:
        br = br - (br - ul - res)
    pad = int(np.ceil(np.linalg.norm(br - ul) / 2 - float(br[1] - ul[1]) / 2))
    if not rot == 0:
        ul = ul - pad
        br = br + pad
    return scale_factor, center, scale, ul, br, pad


def crop(img, center, scale, rot, res, size=200, quirk=True):
    """pylib/HumanAug.py:117-176.  img: H x W x 3 float32 in [0,1] (what im_to_numpy(load_image(...)) hands over after the
    flip and the colour gain of data/mpii_for_mpii.py:126-135).  Returns res x res x 3 uint8."""
    img = np.asarray(img)
    rot = float(np.asarray(rot, dtype=np.float64).reshape(-1)[0])
    scale_factor, center, scale, ul, br, pad = crop_window(center, scale, rot, res, size)
    full = 1.0
    if scale_factor >= 2:
        new_img_size = np.floor(max(img.shape[0], img.shape[1]) / scale_factor)
        if new_img_size < 2:
            return img
        img = imresize_fraction(img, 1 / scale_factor, quirk, 1.0)       # uint8 from here on
        full = 255.0
    new_shape = [br[1] - ul[1], br[0] - ul[0], img.shape[2]]
    new_img = np.zeros(new_shape)
    ht, wd = img.shape[0], img.shape[1]
    new_x = max(0, -ul[0]), min(br[0], wd) - ul[0]
    new_y = max(0, -ul[1]), min(br[1], ht) - ul[1]
    old_x = max(0, ul[0]), min(wd, br[0])
    old_y = max(0, ul[1]), min(ht, br[1])
    new_img[new_y[0]:new_y[1], new_x[0]:new_x[1]] = img[old_y[0]:old_y[1], old_x[0]:old_x[1]]
    if not rot == 0:
        new_img = imrotate(new_img, rot, quirk, full)
        new_img = new_img[pad:-pad, pad:-pad]
    return imresize_shape(new_img, (res, res), quirk, full)


def source_image(frame_u8, flip=False, gain=(1., 1., 1.)):
    """What the dataset hands to crop (data/mpii_for_mpii.py:114-135, utils/imutils.py:31-40): the uint8 frame as fp32 / 255
    (torch float32 arithmetic), mirrored along W, each channel times its gain (the Python float rounded to fp32, product
    in fp32) and clamped to [0, 1].  H x W x 3 float32."""
    img = frame_u8.astype(np.float32) / np.float32(255)
    if flip:
        img = img[:, ::-1]
    g = np.asarray(gain, dtype=np.float32)
    return np.clip(img * g[None, None, :], np.float32(0), np.float32(1)).astype(np.float32)


def crop_frame(frame_u8, center, scale, rot, res, flip=False, gain=(1., 1., 1.), quirk=False):
    """The device warp's contract: uint8 frame + augmentation parameters -> network input 3 x res x res float32 in [0, 1]
    (utils/imutils.py:31-36: uint8 crop / 255).  `center` is the centre AFTER the mirror (c.x = W - c.x is the caller's)."""
    out = crop(source_image(frame_u8, flip, gain), center, scale, rot, res, 200, quirk)
    return np.ascontiguousarray(out.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
