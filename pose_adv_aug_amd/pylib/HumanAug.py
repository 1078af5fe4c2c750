"""pylib/HumanAug.py of the reference, on the GPU: similarity transforms, joint transforms and the
scale/rotation crop (pre-downscale, window, rotate, resize: the reference's own pixel pipeline) on the device."""
from .. import _lib
from ._dev import lib, check, ptr, stream, dev, to_dev, np, torch

FLIP_PAIRS = ((0, 5), (1, 4), (2, 3), (10, 15), (11, 14), (12, 13))          # pylib/HumanAug.py:241-244
FLIP_CHANNELS = ((1, 4), (0, 5), (12, 13), (11, 14), (10, 15), (2, 3))        # pylib/HumanAug.py:182


def _scalar(v):
    return float(np.asarray(v, dtype=np.float64).reshape(-1)[0])


def make_params(center, scale, rot, flip=None, gain=None):
    """[B][8] float64 {cx, cy, scale, rot, flip, gain_r, gain_g, gain_b} (layout of include/poseadv.h)."""
    c = np.asarray(center, dtype=np.float64).reshape(-1, 2)
    B = c.shape[0]
    p = np.zeros((B, 8), dtype=np.float64)
    p[:, 0:2] = c
    p[:, 2] = np.asarray(scale, dtype=np.float64).reshape(-1)
    p[:, 3] = np.asarray(rot, dtype=np.float64).reshape(-1)
    p[:, 4] = 0 if flip is None else np.asarray(flip, dtype=np.float64).reshape(-1)
    p[:, 5:8] = 1 if gain is None else np.asarray(gain, dtype=np.float64).reshape(-1, 3)
    return torch.from_numpy(p).to(dev())


def affine_params(params, res_in=256, res_out=64):
    """device: params [B][8] -> (T at res_out [B][6] float64, T^-1 at res_in [B][6] float64)."""
    B = params.shape[0]
    t = torch.empty((B, 6), dtype=torch.float64, device=dev())
    ti = torch.empty((B, 6), dtype=torch.float64, device=dev())
    check(lib().pa_affine_params(ptr(params), B, res_in, res_out, ptr(t), ptr(ti), stream()), 'pa_affine_params')
    return t, ti


def GetTransform(center, scale, rot, res, size):
    """pylib/HumanAug.py:10-35 -> 3x3 float64 numpy."""
    if size != 200:
        raise ValueError('the reference always uses size == 200')
    p = make_params(np.asarray(center, dtype=np.float64).reshape(1, 2), [_scalar(scale)], [_scalar(rot)])
    t, _ = affine_params(p, res_in=int(res), res_out=int(res))
    out = np.eye(3)
    out[:2, :] = t.cpu().numpy().reshape(2, 3)
    return out


def TransformPts(pts, center, scale, rot, res, size, invert=0):
    """pylib/HumanAug.py:45-54 (0-based, float result)."""
    t = GetTransform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    p = np.concatenate((np.asarray(pts, dtype=np.float64), np.ones((len(pts), 1))), axis=1).T
    return np.dot(t, p)[0:2, :].T


def transform_pts_batch(pts, params, t, width, sizes=None):
    """device: joints [B][J][2] fp32 image px -> (heat-map coords [B][J][2] float64 with invalid joints
    zeroed, mirrored/swapped image-space joints [B][J][2] fp32).  pylib/HumanAug.py:45-54,236-257 and
    data/mpii_for_mpii.py:138-146."""
    p = to_dev(pts, torch.float32)
    B, J = p.shape[0], p.shape[1]
    out = torch.empty((B, J, 2), dtype=torch.float64, device=dev())
    img = torch.empty((B, J, 2), dtype=torch.float32, device=dev())
    if sizes is not None:          # frames of different sizes: every sample mirrors about its own width
        check(lib().pa_transform_pts_sized(ptr(p), ptr(params), ptr(t), B, J, ptr(sizes), ptr(out), ptr(img), stream()),
              'pa_transform_pts_sized')
    else:
        check(lib().pa_transform_pts(ptr(p), ptr(params), ptr(t), B, J, float(width), ptr(out), ptr(img), stream()),
              'pa_transform_pts')
    return out, img


_CROP_WS = {}


def _crop_workspace(B, Hs, Ws, res):
    """scratch of pa_crop (intermediate images of the staged crop): ONE grow-only buffer per (device, stream it is used
    on), re-allocated only when pa_crop_workspace_bytes asks for more than it holds -- the MPII feed pads every batch to its
    own (Hs, Ws), a cache per shape would grow by ~90 MB per new shape.  Keyed by the stream so that the crop enqueued one step
    ahead on the augmentation stream (data.AugmentAhead) and a crop on the main stream (validate) never share scratch; a
    buffer that is outgrown is released through torch's stream-ordered allocator on the stream that used it."""
    need = int(lib().pa_crop_workspace_bytes(B, Hs, Ws, res))
    key = (str(dev()), int(getattr(stream(), 'value', stream()) or 0))
    ws = _CROP_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _CROP_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev())
    return ws


def crop_batch(frames, params, res=256, want_nchw=False, want_nhwc4=True, want_u8=False, sizes=None):
    """device: uint8 frames [B][Hs][Ws][3] + params [B][8] -> network input (bf16 NHWC4) and/or fp32 NCHW and/or the uint8
    crop [B][res][res][3].  crop (pylib/HumanAug.py:117-176: pre-downscale, int-truncated window, PIL rotate, PIL resize)
    behind flip + colour gain (data/mpii_for_mpii.py:126-135), byte-exact against the reference over Pillow."""
    f = frames if (isinstance(frames, torch.Tensor) and frames.device == dev()) else to_dev(frames, torch.uint8)
    B, Hs, Ws, _ = f.shape
    out4 = torch.empty((B, res, res, 4), dtype=_lib.act_dtype(), device=dev()) if want_nhwc4 else None
    outf = torch.empty((B, 3, res, res), dtype=torch.float32, device=dev()) if want_nchw else None
    out8 = torch.empty((B, res, res, 3), dtype=torch.uint8, device=dev()) if want_u8 else None
    check(lib().pa_crop(ptr(f), Hs, Ws, ptr(sizes), ptr(params), B, res, ptr(_crop_workspace(B, Hs, Ws, res)), ptr(out4), ptr(outf),
                        ptr(out8), stream()), 'pa_crop')
    return out4, outf, out8


def warp_batch(frames, tinv, params, res=256, want_nchw=False, want_nhwc4=True, sizes=None):
    """device: the PURE inverse-affine bilinear sampler (2 x 2 taps at T^-1(u, v), no pre-filter) -- an operator, not the
    reference's crop pixels (crop_batch is what the loops use)."""
    f = frames if (isinstance(frames, torch.Tensor) and frames.device == dev()) else to_dev(frames, torch.uint8)
    B, Hs, Ws, _ = f.shape
    out4 = torch.empty((B, res, res, 4), dtype=_lib.act_dtype(), device=dev()) if want_nhwc4 else None
    outf = torch.empty((B, 3, res, res), dtype=torch.float32, device=dev()) if want_nchw else None
    if sizes is not None:
        check(lib().pa_affine_warp_bilinear_sized(ptr(f), Hs, Ws, ptr(sizes), ptr(tinv), ptr(params), B, res, ptr(out4), ptr(outf),
                                                  stream()), 'pa_affine_warp_bilinear_sized')
    else:
        check(lib().pa_affine_warp_bilinear(ptr(f), Hs, Ws, ptr(tinv), ptr(params), B, res, ptr(out4), ptr(outf), stream()),
              'pa_affine_warp_bilinear')
    return out4, outf


def crop(img, center, scale, rot, res, size):
    """Reference signature (pylib/HumanAug.py:117): H x W x 3 image (float [0,1] or uint8) -> res x res x 3 uint8, with the
    reference's pixels (its PIL arithmetic restated on the device) except scipy's per-image min/max byte stretching."""
    if size != 200:
        raise ValueError('the reference always uses size == 200')
    a = np.asarray(img)
    if a.dtype != np.uint8:
        a = (np.clip(a.astype(np.float64) * 255.0, 0, 255) + 0.5).astype(np.uint8)      # toimage() with the full range
    p = make_params(np.asarray(center, dtype=np.float64).reshape(1, 2), [_scalar(scale)], [_scalar(rot)])
    _, _, out8 = crop_batch(np.ascontiguousarray(a)[None], p, res=int(res), want_nhwc4=False, want_u8=True)
    return out8[0].cpu().numpy()


def shufflelr(x, width, dataset='mpii'):
    """pylib/HumanAug.py:236-257 (in place on a 16 x 2 tensor, like the reference)."""
    assert dataset == 'mpii'
    x[:, 0] = width - x[:, 0]
    for a, b in FLIP_PAIRS:
        tmp = x[a, :].clone(); x[a, :] = x[b, :]; x[b, :] = tmp
    return x


def fliplr(x):
    """pylib/HumanAug.py:260-266: mirror the last axis of a CHW / NCHW numpy array."""
    return np.ascontiguousarray(np.asarray(x)[..., ::-1]).astype(float)


def flip_channels(maps):
    """pylib/HumanAug.py:198-210."""
    return maps.flip(-1).float()


def flip_lr_img4(img4):
    """device: mirror the bf16 NHWC4 network input along W (stack-hg.py:223 `img.numpy()[:, :, :, ::-1]`)."""
    B, H, W, _ = img4.shape
    out = torch.empty_like(img4)
    check(lib().pa_flip_lr_nhwc4(ptr(img4), ptr(out), B, H, W, stream()), 'pa_flip_lr_nhwc4')
    return out


def flip_tta_merge(output, output_flipped):
    """device, one kernel: (output + shuffle_channels_for_horizontal_flipping(flip_channels(output_flipped))) / 2
    (stack-hg.py:228-230) for NCHW fp32 MPII heat maps."""
    a = to_dev(output, torch.float32); b = to_dev(output_flipped, torch.float32)
    B, J, H, W = a.shape
    out = torch.empty_like(a)
    check(lib().pa_flip_tta_merge(ptr(a), ptr(b), ptr(out), B, J, H, W, stream()), 'pa_flip_tta_merge')
    return out


def shuffle_channels_for_horizontal_flipping(maps):
    """pylib/HumanAug.py:179-196 (in place)."""
    dim = 1 if maps.dim() == 4 else 0
    for a, b in FLIP_CHANNELS:
        tmp = maps.narrow(dim, a, 1).clone()
        maps.narrow(dim, a, 1).copy_(maps.narrow(dim, b, 1))
        maps.narrow(dim, b, 1).copy_(tmp)
    return maps
