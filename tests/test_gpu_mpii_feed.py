"""Real-data feed: MPII-format JSON + image files -> DeviceBatch with per-sample frame sizes."""
import json

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
