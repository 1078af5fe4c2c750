"""Real-data feed: MPII-format JSON + image files -> DeviceBatch with per-sample frame sizes."""
import json
import os

import numpy as np
import pytest
import torch

from tests import inputs

pytestmark = pytest.mark.gpu


def _make_dataset(tmp_path, n=7):
    from PIL import Image
    g = inputs.rng(900)
    sizes = [(320, 240), (200, 260), (256, 256), (400, 180), (300, 300), (180, 220), (350, 210)]
    anno = []
    for i in range(n):
        w, h = sizes[i % len(sizes)]
        img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(img).save(str(tmp_path / ('im%d.png' % i)))             # lossless: exact pixel comparisons
        joints = np.concatenate([g.uniform(5, [w - 5, h - 5], (16, 2)), np.ones((16, 1))], 1)
        joints[g.random(16) < 0.15, 0:2] = 0
        anno.append({'dataset': 'MPII', 'isValidation': float(i % 3 == 0), 'img_paths': 'im%d.png' % i,
                     'joint_self': joints.tolist(), 'objpos': [w / 2.0 + float(g.uniform(-10, 10)), h / 2.0 + float(g.uniform(-10, 10))],
                     'scale_provided': float(g.uniform(0.8, 1.4)), 'normalizer': float(g.uniform(40, 90))})
    anno.append({'dataset': 'LEEDS', 'isValidation': 0.0, 'img_paths': 'x.png', 'joint_self': [[0, 0, 0]] * 16,
                 'objpos': [1, 1], 'scale_provided': 1.0, 'normalizer': 1.0})   # other datasets are skipped (:34)
    path = tmp_path / 'anno.json'
    path.write_text(json.dumps(anno))
    return str(path), anno, sizes


def test_json_split_prenormalisation_and_padded_frames(tmp_path):
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    path, anno, sizes = _make_dataset(tmp_path)
    lines = []
    tr = MPII(path, str(tmp_path), is_train=True, log=lines.append)
    va = MPII(path, str(tmp_path), is_train=False, log=lines.append)
    assert tr.train == [1, 2, 4, 5] and tr.valid == [0, 3, 6] and len(tr) == 4 and len(va) == 3
    assert lines[0] == 'loading json file is done...' and lines[1] == 'total training images: 4'
    b = va.load_batch([0, 1, 2])
    assert b.index == [0, 1, 2] and b.B == 3
    for k, ai in enumerate([0, 3, 6]):
        a = anno[ai]
        w, h = sizes[ai % len(sizes)]
        assert tuple(b.sizes[k].tolist()) == (w, h)
        s = np.float32(a['scale_provided'])
        assert abs(float(b.meta[k, 1]) - (np.float32(a['objpos'][1]) + 15 * s)) < 1e-4          # c.y += 15 s
        assert abs(float(b.meta[k, 2]) - float(s * np.float32(1.25))) < 1e-6                     # s *= 1.25
        assert abs(float(b.normalizer[k]) - float(np.float32(a['normalizer']) * np.float32(0.6))) < 1e-5
        assert float(b.meta[k, 3]) == w                                                          # own width
        from PIL import Image
        ref = np.asarray(Image.open(str(tmp_path / a['img_paths'])).convert('RGB'))
        fr = b.frames[k].cpu().numpy()
        assert np.array_equal(fr[:h, :w], ref) and not fr[h:].any() and not fr[:, w:].any()
    got = [bb.index for bb in tr.batches(3, shuffle=False)]
    assert got == [[0, 1, 2], [3]]
    assert sorted(sum([bb.index for bb in tr.batches(3, seed=1)], [])) == [0, 1, 2, 3]


def test_padded_batch_equals_single_image_batches(tmp_path):
    """The warp and the joint transform of a padded multi-size batch (flip included: mirror about the sample's OWN width)
    equal those of every sample processed alone with its exact frame."""
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    path, anno, sizes = _make_dataset(tmp_path)
    ds = MPII(path, str(tmp_path), is_train=True, log=lambda *_: None)
    batch = ds.load_batch([0, 1, 2, 3])
    aug = Augmenter(seed=11)
    data = aug.regular(batch)
    params = batch.params.clone()
    assert float(params[:, 4].sum()) > 0, 'seed should flip at least one sample'
    for k in range(batch.B):
        w, h = batch.sizes[k].tolist()
        single = DeviceBatch(batch.frames[k:k + 1, :h, :w].contiguous(), batch.meta[k:k + 1, 0:2].cpu(), batch.meta[k:k + 1, 2].cpu(),
                             batch.joints[k:k + 1].cpu(), batch.normalizer[k:k + 1].cpu())
        single.params.copy_(params[k:k + 1])
        d1 = aug._finish(single)
        assert torch.equal(d1['img4'][0], data['img4'][k]), k
        assert torch.equal(d1['pts'][0], data['pts'][k]) and torch.equal(d1['grnd_pts'][0], data['grnd_pts'][k]), k


def test_process_pool_feed_equals_thread_feed_and_drives_train(tmp_path):
    """decoder='process' (forked workers decoding into shared page-locked frame slots, several batches in flight, copies on
    their own stream) yields the same batches as the thread feeder -- indices, sizes, annotations, and every pixel inside each
    sample's own frame -- over two passes (slots are reused), and stack_hg.train / validate run from the sized feed
    (len(), enumerate, num_samples) like the reference's loops from their DataLoader (stack-hg.py:133,183)."""
    import types
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter
    from pose_adv_aug_amd import stack_hg
    path, anno, sizes = _make_dataset(tmp_path, n=7)
    ds = MPII(path, str(tmp_path), is_train=True, log=lambda *_: None)
    ft = ds.batches(2, shuffle=False, decoder='thread')
    fp = ds.batches(2, shuffle=False, decoder='process', workers=3, prefetch=2)
    assert len(ft) == len(fp) == 2 and fp.num_samples == 4
    for _ in range(2):
        for a, b in zip(ft, fp):
            assert a.index == b.index and torch.equal(a.sizes, b.sizes) and torch.equal(a.meta, b.meta)
            assert torch.equal(a.joints, b.joints) and torch.equal(a.normalizer, b.normalizer)
            for k in range(a.B):
                w, h = a.sizes[k].tolist()
                assert torch.equal(a.frames[k, :h, :w], b.frames[k, :h, :w]), (a.index, k)
    net = create_hg(1, 1, 16, 128, default_batch=2); net.reset_parameters(seed=0)
    opt = RMSprop(net, lr=2.5e-4)
    o = types.SimpleNamespace(print_freq=1)
    lines = []
    tl, tp = stack_hg.train(fp, net, opt, Augmenter(seed=1), 0, o, log=lines.append)
    assert len(lines) == 2 and lines[0].startswith('epoch:0, iters:0/2 loss: ') and np.isfinite(tl)
    va = MPII(path, str(tmp_path), is_train=False, log=lambda *_: None).batches(2, decoder='process', workers=2, prefetch=2)
    vl, vp, preds = stack_hg.validate(va, net, Augmenter(seed=2), 0, o, log=lambda m: None)
    assert preds.shape == (3, 16, 2) and np.isfinite(vl)


def test_abandoned_passes_return_their_frame_slots(tmp_path):
    """joint-train...:180-191 takes ONE batch of a fresh pass per epoch (next(iter(feed))): every abandoned pass must hand its
    frame slots back, or the feed runs dry after a few epochs"""
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    path, anno, sizes = _make_dataset(tmp_path, n=7)
    ds = MPII(path, str(tmp_path), is_train=True, log=lambda *_: None)
    feed = ds.batches(1, shuffle=False, decoder='process', workers=2, prefetch=2)
    first = [next(iter(feed)).index for _ in range(12)]          # 12 abandoned passes with 4 slots
    assert first == [[0]] * 12
    assert [b.index for b in feed] == [[0], [1], [2], [3]]


def _engine_sample(ds, index, mode, draws7, si=None, ri=None):
    """one person through the product path: JSON + PNG -> MPII.load_batch -> law with the given draws -> device crop, joints,
    heat maps.  Returns (crop bytes [3][256][256], heat [16][64][64], c, s, r, grnd pts, normalizer, params row)."""
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    from pose_adv_aug_amd.data import Augmenter
    from pose_adv_aug_amd.pylib import HumanAug, HumanPts
    batch = ds.load_batch([index])
    aug = Augmenter(seed=0)
    if mode is None:                                   # validation: the un-augmented crop
        data = aug.standard(batch)
    else:
        dd = torch.from_numpy(np.asarray(draws7, dtype=np.float64).reshape(1, 7)).cuda()
        dsi = None if si is None else torch.tensor([si], dtype=torch.int32).cuda()
        dri = None if ri is None else torch.tensor([ri], dtype=torch.int32).cuda()
        check(lib().pa_sample_aug_given(ptr(batch.meta), ptr(dsi), ptr(dri), mode, ptr(dd), 1, ptr(batch.params), stream()))
        data = aug._finish(batch)
    _, _, u8 = HumanAug.crop_batch(batch.frames, batch.params, res=256, want_nhwc4=False, want_u8=True, sizes=batch.sizes)
    heat = HumanPts.pts2heatmap_batch(data['pts'], 64, 64)
    return (u8[0].permute(2, 0, 1).cpu().numpy(), heat[0].cpu().numpy(), data['c'][0].cpu().numpy(), float(data['s'][0]), float(data['r'][0]),
            data['grnd_pts'][0].cpu().numpy(), float(data['normalizer'][0]), data['img4'][0].float().cpu().numpy())


def test_whole_samples_equal_the_reference_dataset_golden(tmp_path):
    """tests/golden/dataset.npz holds what the REFERENCE's MPII.__getitem__ / AGENT.__getitem__ return for seeded np.random
    draws (data/mpii_for_mpii.py:83-163, data/joint_train_s_r_agent.py:98-177; tests/golden/make_goldens.py gen_dataset).  The
    product path -- mpii_for_mpii.MPII from the same JSON + PNG files, the same draws through pa_sample_aug_given, pa_crop,
    pa_transform_pts, the Gaussian maps -- gives the same 7-tuple: heat maps bit for bit, c / s / r / joints / normaliser exactly,
    crop bytes exactly whenever scipy's per-image byte stretch is the identity (a black and a white pixel inside the crop:
    SURVEY.md Appendix A.13; otherwise the device equals the oracle without that accident, which test_oracle_golden pins)."""
    from oracle import pylib as opl
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dataset.npz'))
    path, frames, anno = inputs.write_dataset(str(tmp_path))
    tr = MPII(path, str(tmp_path), is_train=True, log=lambda *_: None)
    va = MPII(path, str(tmp_path), is_train=False, log=lambda *_: None)
    exact_crops = 0

    def check_sample(tag, got, frame, oracle_sample):
        nonlocal exact_crops
        u8, heat, c, s, r, pts, normalizer, img4 = got
        assert np.array_equal(heat, G[tag + '_heat']), tag                                   # heat maps bit for bit
        assert np.array_equal(c.astype(np.float32), G[tag + '_c']) and np.float32(s) == G[tag + '_s'].reshape(-1)[0], tag
        assert np.float32(r) == G[tag + '_r'].reshape(-1)[0] and np.array_equal(pts.astype(np.float32), G[tag + '_pts']), tag
        assert abs(normalizer - float(G[tag + '_normalizer'])) <= 1e-6 * float(G[tag + '_normalizer']), tag
        plain = np.rint(oracle_sample(False)[0] * 255).astype(np.uint8)                     # the oracle without scipy's stretch
        assert np.array_equal(u8, plain), tag                                               # ... is what the device computes, byte for byte
        sums = np.array([u8[k].astype(np.int64).sum() for k in range(3)] + [(u8[k].astype(np.int64) ** 2).sum() for k in range(3)])
        if np.array_equal(u8[:, 1::4, 2::4], G[tag + '_inp_sub']) and np.array_equal(sums, G[tag + '_inp_sums']):
            exact_crops += 1                                                                # equals the reference's own bytes
        else:                                                                               # only where the stretch was NOT the identity
            stretched = np.rint(oracle_sample(True)[0] * 255).astype(np.uint8)
            assert not np.array_equal(stretched, plain), tag

    n = 0
    for index in (0, 1):
        for seed in (11, 12, 13, 14, 15, 16):
            d = inputs.legacy_draws(seed)
            check_sample('train%d_seed%d' % (index, seed), _engine_sample(tr, index, 0, d), frames[index],
                         lambda q, i=index, d=d: opl.mpii_getitem(frames[i], anno[i], d, True, quirk=q))
            n += 1
    check_sample('val0', _engine_sample(va, 0, None, None), frames[2], lambda q: opl.mpii_getitem(frames[2], anno[2], None, False, quirk=q))
    for k in (0, 1):
        st = np.random.RandomState(21 + k)
        a = np.array([st.randn(), st.randn()] + [st.random_sample() for _ in range(4)])
        d7 = np.array([a[0], a[1], 0.5, a[2], a[3], a[4], a[5]])                             # (slot 2, "rotation forced to 0", is not drawn by the agent's law)
        i, si, ri = int(G['agent_img_index'][k]), int(G['agent_scale_index'][k]), int(G['agent_rot_index'][k])
        check_sample('agent%d' % k, _engine_sample(tr, i, 1, d7, si, ri), frames[i],
                     lambda q, i=i, si=si, ri=ri, a=a: opl.agent_getitem(frames[i], anno[i], si, ri, a, quirk=q))
    st = np.random.RandomState(31)
    a = np.array([st.randn(), st.randn(), 0, 0, 0, 0])
    d7 = np.array([a[0], a[1], 0.5, 0.9, 0.5, 0.5, 0.5])
    for mode, name, k in ((2, 'sep_scale', 0), (3, 'sep_rot', 1)):
        check_sample(name, _engine_sample(tr, 0, mode, d7, 6, 2), frames[0],
                     lambda q, k=k, a=a: opl.agent_getitem(frames[0], anno[0], 6, 2, a, separate_s_r=True, quirk=q)[k])
    assert exact_crops >= 8, exact_crops                   # most crops carry a black and a white pixel: the reference's own bytes
