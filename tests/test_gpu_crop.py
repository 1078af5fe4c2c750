"""Row a9 / a10 on the device: the staged crop (pa_crop) against the REFERENCE's crop (tests/golden/crop.npz: outputs of
pylib/HumanAug.py:crop over the real Pillow, made by tests/golden/make_goldens.py) and against the oracle at the benchmark's
own workload; the augmentation laws value for value against the oracle with caller-given draws."""
import os

import numpy as np
import pytest
import torch

from tests import inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def P():
    import pose_adv_aug_amd.pylib as P
    return P


def _case_params(P, cases):
    c = np.array([k[1] for k in cases], dtype=np.float32)
    flip = np.array([k[4] for k in cases])
    c[:, 0] = np.where(flip == 1, np.float32(1280) - c[:, 0], c[:, 0])          # the caller mirrors the centre (:129)
    s = np.array([k[2] for k in cases], dtype=np.float32)
    return P.HumanAug.make_params(c.astype(np.float64), s.astype(np.float64), [k[3] for k in cases], flip=flip,
                                  gain=np.array([k[5] for k in cases], dtype=np.float64))


def test_device_crop_equals_the_reference_crop(P):
    """720x1280 frames, res 256, the survey's (s, r) set and more (pre-downscale on / off, rotation, mirror, colour gain,
    window outside the frame): BYTE-exact against the reference's own output.  Bound asserted: every pixel equal."""
    g = np.load(os.path.join(G, 'crop.npz'))
    cases = inputs.WARP_CASES
    frames = {k: inputs.warp_frame(k) for k in sorted({c[0] for c in cases})}
    batch = np.stack([frames[c[0]] for c in cases])
    params = _case_params(P, cases)
    out4, outf, out8 = P.HumanAug.crop_batch(batch, params, res=256, want_nchw=True, want_u8=True)
    out8 = out8.cpu().numpy()
    for i, case in enumerate(cases):
        if not case[6]:
            continue                                   # byte-stretching quirk cases: see the next test
        sub = g['crop%02d_sub' % i]
        d = np.abs(out8[i][1::4, 2::4].astype(int) - sub.astype(int))
        assert d.max() == 0, (i, case, int(d.max()), float((d > 0).mean()))
        sums = [int(out8[i][..., k].astype(np.int64).sum()) for k in range(3)] + \
               [int((out8[i][..., k].astype(np.int64) ** 2).sum()) for k in range(3)]
        assert sums == [int(v) for v in g['crop%02d_sums' % i]], (i, case)
        if 'crop%02d_full' % i in g.files:
            assert np.array_equal(out8[i], g['crop%02d_full' % i])
    # network layouts: uint8 / 255 (utils/imutils.py:31-36), bf16 NHWC4 with a zero 4th channel
    ref = torch.from_numpy(out8).float() / 255
    assert torch.equal(outf.cpu(), ref.permute(0, 3, 1, 2))
    o4 = out4.float().cpu()
    assert torch.equal(o4[..., :3], ref.to(torch.bfloat16).float()) and float(o4[..., 3].abs().max()) == 0


def test_device_crop_without_the_byte_stretching_quirk(P):
    """Crops with no pure black / white pixel: scipy's toimage() stretches [min, max] to [0, 255]; the device does not
     and equals oracle.crop(quirk=False), whose quirk=True twin reproduces the reference."""
    from oracle import crop as oc
    cases = [c for c in inputs.WARP_CASES if not c[6]]
    frames = {k: inputs.warp_frame(k) for k in sorted({c[0] for c in cases})}
    params = _case_params(P, cases)
    _, _, out8 = P.HumanAug.crop_batch(np.stack([frames[c[0]] for c in cases]), params, res=256, want_nhwc4=False, want_u8=True)
    for i, (kind, c0, s0, r0, flip, gain, _) in enumerate(cases):
        ref = oc.crop(oc.source_image(frames[kind], flip, gain), np.array(c0, dtype=np.float32), np.float32(s0), r0, 256, 200, quirk=False)
        assert np.array_equal(out8[i].cpu().numpy(), ref), (i, kind, s0, r0)


def test_benchmark_workload_crops_equal_the_oracle(P):
    """BASELINE configs[1]'s own input: the synthetic 720x1280 uint8 people of DeviceBatch.synthetic(24), parameters drawn
    by the regular law on the device -> every one of the 24 crops equals oracle.crop_frame byte for byte."""
    from oracle import crop as oc
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    batch = DeviceBatch.synthetic(24, seed=3)
    aug = Augmenter(seed=11)
    aug.regular(batch)                                           # draws batch.params
    p = batch.params.cpu().numpy()
    assert ((p[:, 2] * 200 / 256) >= 2).any() and ((p[:, 2] * 200 / 256) < 2).any() and (p[:, 3] != 0).any()
    _, outf, out8 = P.HumanAug.crop_batch(batch.frames, batch.params, res=256, want_nchw=True, want_nhwc4=False, want_u8=True)
    frames = batch.frames.cpu().numpy()
    worst = 0
    for i in range(24):
        ref = oc.crop_frame(frames[i], p[i, 0:2], p[i, 2], p[i, 3], 256, flip=bool(p[i, 4]), gain=p[i, 5:8], quirk=False)
        d = np.abs(outf[i].cpu().numpy() - ref)
        worst = max(worst, float(d.max()))
        assert d.max() == 0, (i, p[i], float(d.max()) * 255, float((d > 0).mean()))
    assert worst == 0


def test_sized_frames_crop_like_single_frames(P):
    """frames of different sizes in one padded buffer (real MPII images): each sample crops exactly like its own frame"""
    rng = inputs.rng(95)
    sizes = np.array([[640, 480], [500, 375], [1280, 720]], dtype=np.int32)
    Hs, Ws = 720, 1280
    buf = np.zeros((3, Hs, Ws, 3), dtype=np.uint8)
    singles = []
    for i, (w, h) in enumerate(sizes):
        f = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        buf[i, :h, :w] = f
        singles.append(f)
    c = np.array([[300.0, 250.0], [260.5, 180.25], [655.0, 371.0]])
    s = np.array([1.7, 2.9, 3.3]); r = np.array([12.0, 0.0, -33.0]); flip = [1, 0, 1]
    gain = np.array([[1.1, 0.8, 1.0], [1, 1, 1], [0.65, 1.3, 1.2]])
    params = P.HumanAug.make_params(c, s.astype(np.float32).astype(np.float64), r, flip=flip, gain=gain)
    _, _, out = P.HumanAug.crop_batch(buf, params, res=256, want_nhwc4=False, want_u8=True, sizes=torch.from_numpy(sizes).cuda())
    for i in range(3):
        _, _, one = P.HumanAug.crop_batch(np.ascontiguousarray(singles[i])[None], params[i:i + 1].contiguous(), res=256, want_nhwc4=False, want_u8=True)
        assert torch.equal(out[i], one[0]), i


def test_reference_signature_crop(P):
    img = inputs.warp_frame('smooth').astype(np.float64) / 255.0
    from oracle import crop as oc
    cr = P.HumanAug.crop(img, [655.3, 371.8], np.float32(2.013), 25.0, 256, 200)
    ref = oc.crop(oc.source_image(inputs.warp_frame('smooth')), np.array([655.3, 371.8], dtype=np.float32), np.float32(2.013), 25.0, 256, 200, quirk=False)
    assert cr.shape == (256, 256, 3) and cr.dtype == np.uint8 and np.array_equal(cr, ref)


def test_augmentation_laws_value_for_value(P):
    """pa_sample_aug_given with the draws of a numpy generator in the reference's call order == oracle.regular_aug /
    agent_aug (data/mpii_for_mpii.py:119-135, data/joint_train_s_r_agent.py:134-169): centre, scale, rotation, flip, gains."""
    from oracle import pylib as opl
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    B = 512
    g = inputs.rng(96)
    meta = np.stack([g.uniform(300, 900, B), g.uniform(200, 500, B), g.uniform(1.5, 3.5, B) * 1.25, np.full(B, 1280.0)], 1).astype(np.float32)
    draws = np.concatenate([g.standard_normal((B, 2)) * np.array([[1.5, 1.5]]), g.random((B, 5))], 1)     # wide normals: both clips fire
    draws[0, 2] = 0.6; draws[1, 3] = 0.5                                                                     # the <= boundaries
    si = g.integers(0, 7, B).astype(np.int32); ri = g.integers(0, 7, B).astype(np.int32)
    dm, dd = torch.from_numpy(meta).cuda(), torch.from_numpy(draws).cuda()
    dsi, dri = torch.from_numpy(si).cuda(), torch.from_numpy(ri).cuda()
    out = torch.zeros(B, 8, dtype=torch.float64, device='cuda')
    check(lib().pa_sample_aug_given(ptr(dm), None, None, 0, ptr(dd), B, ptr(out), stream()))
    p = out.cpu().numpy()
    for b in range(B):
        c, s, r, flip, gains = opl.regular_aug(meta[b, 0:2], meta[b, 2], 1280.0, draws[b, 0], draws[b, 1], draws[b, 2], draws[b, 3], draws[b, 4:7])
        assert np.array_equal(p[b, 0:2], c) and p[b, 2] == s and p[b, 3] == r and bool(p[b, 4]) == flip and np.array_equal(p[b, 5:8], gains), b
    assert (p[:, 3] == 0).any() and (np.abs(p[:, 3]) == 60).any() and (p[:, 4] == 1).any()
    check(lib().pa_sample_aug_given(ptr(dm), ptr(dsi), ptr(dri), 1, ptr(dd), B, ptr(out), stream()))
    p = out.cpu().numpy()
    for b in range(B):
        c, s, r, flip, gains = opl.agent_aug(meta[b, 0:2], meta[b, 2], 1280.0, si[b], ri[b], draws[b, 0], draws[b, 1], draws[b, 3], draws[b, 4:7])
        assert np.array_equal(p[b, 0:2], c) and p[b, 2] == s and p[b, 3] == r and bool(p[b, 4]) == flip and np.array_equal(p[b, 5:8], gains), b
    # scale-only / rotation-only agent crops (separate_s_r, data/joint_train_s_r_agent.py:140-160): no flip, no gain
    check(lib().pa_sample_aug_given(ptr(dm), ptr(dsi), ptr(dri), 2, ptr(dd), B, ptr(out), stream()))
    p2 = out.cpu().numpy()
    check(lib().pa_sample_aug_given(ptr(dm), ptr(dsi), ptr(dri), 3, ptr(dd), B, ptr(out), stream()))
    p3 = out.cpu().numpy()
    assert np.array_equal(p2[:, 2], p[:, 2]) and np.all(p2[:, 3] == 0) and np.all(p2[:, 4] == 0) and np.all(p2[:, 5:] == 1)
    assert np.array_equal(p3[:, 3], p[:, 3]) and np.array_equal(p3[:, 2], meta[:, 2].astype(np.float64)) and np.all(p3[:, 4] == 0)


@pytest.mark.parametrize('res', [64, 128, 256, 384])
def test_crop_edge_cases_equal_the_oracle(P, res):
    """Code paths the benchmark's laws never reach, byte for byte against oracle.crop (quirk off): other output resolutions,
    scale factors above 11 (coefficients computed in the passes instead of the tables), heavy up-scaling, windows partly and
    completely outside the frame, a tiny ragged frame, a single sample, rotations by 180 / 90 / 360 degrees."""
    from oracle import crop as oc
    g = inputs.rng(98, res)
    big = g.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    tiny = g.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    k = res / 256.0
    cases = [(big, (210.3, 150.7), 1.7 * k, 0.0), (big, (210.3, 150.7), 14.5 * k, 21.0), (big, (200.0, 140.0), 0.31 * k, -33.0),
             (big, (5.5, 8.25), 1.1 * k, 10.0), (big, (-400.0, -300.0), 0.9 * k, 0.0), (big, (419.0, 299.0), 3.3 * k, 180.0),
             (big, (180.0, 160.0), 2.7 * k, 90.0), (big, (222.0, 111.0), 1.3 * k, 360.0), (tiny, (26.0, 18.0), 0.21 * k, 15.0),
             (tiny, (20.0, 20.0), 4.1 * k, 0.0)]
    for frame, c, s, r in cases:
        s32 = float(np.float32(s))
        params = P.HumanAug.make_params(np.array([c], dtype=np.float32).astype(np.float64), [s32], [r])
        _, _, out8 = P.HumanAug.crop_batch(np.ascontiguousarray(frame)[None], params, res=res, want_nhwc4=False, want_u8=True)
        got = out8[0].cpu().numpy()
        if c[0] < -300:                           # window completely outside: the reference's slice assignment raises there; the device pads
            with pytest.raises(ValueError):
                oc.crop(oc.source_image(frame), np.array(c, dtype=np.float32), np.float32(s), r, res, 200, quirk=False)
            assert got.shape == (res, res, 3) and not got.any()
            continue
        want = oc.crop(oc.source_image(frame), np.array(c, dtype=np.float32), np.float32(s), r, res, 200, quirk=False)
        assert got.shape == want.shape and np.array_equal(got, want), (res, c, s, r, int(np.abs(got.astype(int) - want.astype(int)).max()),
                                                                      float((got != want).mean()))
