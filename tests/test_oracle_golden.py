"""Pins the CPU oracle (oracle/) against golden vectors produced by the
reference itself (tests/golden/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import pylib as opl
from oracle import step as ostep
from tests import inputs

G = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def digest(tensors):
    return np.array([[float(x.sum()), float(x.norm()), float(x.flatten()[0]), float(x.flatten()[-1])]
                     for x in tensors], dtype=np.float64)


# ------------------------------------------------------------------ pylib
def test_heatmap_special_and_random():
    g = load('pylib.npz')
    hm, valid = opl.pts2heatmap(g['hm_special_pts'].copy(), [64, 64])
    assert np.array_equal(hm.astype(np.float32), g['hm_special'])
    assert np.array_equal(valid, g['hm_special_valid'])
    # Appendix A.1 facts
    assert hm[0].argmax() == 3 * 64 + 3 and (hm[1] > 0).sum() == 16 and hm[3].sum() == 0
    pts = inputs.heat_pts(11, 4)
    assert np.array_equal(pts, g['hm_pts'])
    hms = np.stack([opl.pts2heatmap(pts[i].copy(), [64, 64])[0] for i in range(4)])
    assert np.array_equal(hms[:2].astype(np.float32), g['hm_rand'])
    dg = np.array([[h.sum(), (h * h).sum(), h.max()] for h in hms.reshape(-1, 64, 64)])
    assert np.allclose(dg, g['hm_rand_digest'], rtol=0, atol=1e-12)


def test_transforms():
    g = load('pylib.npz')
    c, s, r, pts = g['tf_c'], g['tf_s'], g['tf_r'], g['tf_pts']
    c2, s2, r2, p2, n2 = inputs.person_meta(12, 6)
    assert np.array_equal(c, c2) and np.array_equal(pts, p2) and np.array_equal(g['tf_norm'], n2)
    for i in range(6):
        assert np.array_equal(opl.get_transform(c[i], s[i], r[i], 256), g['tf_T256'][i])
        assert np.array_equal(opl.get_transform(c[i], s[i], r[i], 64), g['tf_T64'][i])
        p64 = opl.transform_pts(pts[i], c[i], s[i], r[i], 64)
        assert np.array_equal(p64, g['tf_pts64'][i])
        back = opl.transform_pts_eval(p64 + 1, c[i], s[i], r[i], 64, invert=1)
        assert np.array_equal(back, g['tf_pts64_eval_inv'][i])
        assert np.array_equal(opl.shufflelr(pts[i], 1280), g['tf_shufflelr'][i])


def _eval_inputs(g):
    c, s, r, gpts, norm = g['tf_c'], g['tf_s'], g['tf_r'], g['tf_pts'], g['tf_norm']
    tp = g['tf_pts64'].copy()
    tp[gpts[..., 0] <= 0] = 0
    tgt = inputs.heatmaps_from_pts(tp, 64)
    pred = inputs.noisy_heatmaps(13, tgt, noise=0.2)
    pred[0, 3] = -1.0
    assert np.allclose([pred.astype(np.float64).sum(), tgt.astype(np.float64).sum()], g['ev_pred_sum'], atol=1e-6)
    return c, s, r, gpts, norm, tgt, pred


def test_evaluation_and_humanacc():
    g = load('pylib.npz')
    c, s, r, gpts, norm, tgt, pred = _eval_inputs(g)
    n = 6
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]
    assert np.array_equal(opl.get_preds(t(pred)).numpy(), g['ev_get_preds'])
    assert np.allclose(opl.accuracy(t(pred), t(tgt), idx).numpy(), g['ev_accuracy'], atol=1e-7)
    cT, sT, rT = t(c).float(), t(s).float().view(n, 1), t(r).float().view(n, 1)
    assert np.array_equal(opl.final_preds(t(pred), cT, sT, [64, 64], rT).numpy(), g['ev_final_preds'])
    a = opl.accuracy_origin_res(t(pred), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)
    assert np.allclose(a.numpy(), g['ev_acc_origin'], atol=1e-7)
    pp = opl.per_person_pckh(t(pred), t(tgt), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)
    assert np.allclose(pp.numpy(), g['ev_per_person'], atol=1e-7)
    pk = opl.approx_pckh(opl.get_preds(t(pred)), opl.get_preds(t(tgt)), idx, 64)
    assert abs(pk - float(g['acc_approx_pckh'])) < 1e-7
    assert np.array_equal(opl.flip_heatmaps(t(pred[:1].copy())).numpy(), g['flip_maps'])
    # the rest of the HumanAcc family (pylib/HumanAcc.py:46-308)
    pp_, gp_ = opl.get_preds(t(pred)), opl.get_preds(t(tgt))
    avg, per = opl.approx_pckh_per(pp_, gp_, idx, 64)
    assert abs(avg - float(g['acc_per_avg'])) < 1e-6 and np.allclose(per.numpy(), g['acc_per'], atol=1e-7)
    _, parts, avg_all = opl.pckh_report(pp_, gp_, t(norm).float() / 20)
    assert np.allclose(np.append(parts.numpy(), avg_all), g['acc_pckh_print'], atol=5.1e-5)       # printed with %.4f
    assert np.array_equal(opl.approx_pckh_samples(pp_, gp_, 64).numpy(), g['acc_samples'])
    assert np.array_equal(opl.correct_predicted_joints(pp_, gp_, 64).numpy(), g['acc_correct'])
    assert np.array_equal(opl.correct_predicted_joints_original_resolution(pp_, gp_, 2.5).numpy(), g['acc_correct_orig'])
    assert np.allclose(opl.predicted_joints_dist_to_grnd(pp_, gp_, 64).numpy(), g['acc_dist_to_grnd'], atol=1e-6)


def test_losses_and_reward_shaping():
    g = load('pylib.npz')
    _, _, _, _, _, tgt, pred = _eval_inputs(g)
    w = inputs.rng(14).random(pred.shape, dtype=np.float32) + 0.5
    assert abs(float(opl.weighted_l2(t(pred), t(tgt), t(w))) - float(g['l2_weighted'])) < 1e-7
    assert abs(float(opl.weighted_l2(t(pred), t(tgt), torch.ones(1))) - float(g['l2_unit'])) < 1e-7
    assert abs(float(opl.stack_mse([t(pred)], t(tgt))) - float(g['l2_unit'])) < 1e-7
    out = opl.gen_groundtruth(t(g['gg_p']), t(g['gg_idx']), t(g['gg_reg']), t(g['gg_agent']))
    assert np.allclose(out.numpy(), g['gg_out'], atol=1e-7)


# ------------------------------------------------------------------- nets
def test_residual_block():
    g = load('residual.npz')
    blk = om.Residual(32, 32)
    om.deterministic_fill_(blk, seed=21)
    blk.train()
    x = t(inputs.rng(22).standard_normal((2, 32, 8, 8)).astype(np.float32)).requires_grad_(True)
    y = blk(x)
    y.backward(t(inputs.rng(23).standard_normal((2, 32, 8, 8)).astype(np.float32)))
    assert np.allclose(y.detach().numpy(), g['y'], atol=1e-6)
    assert np.allclose(x.grad.numpy(), g['dx'], atol=1e-6)
    assert np.allclose(np.concatenate([p.grad.flatten().numpy() for p in blk.parameters()]), g['grads'], atol=1e-5)
    assert np.allclose(np.concatenate([b.flatten().float().numpy() for b in blk.buffers()]), g['running'], atol=1e-6)


@pytest.mark.parametrize('tag,stacks,chan,seed', [('hg_s1c8', 1, 8, 31), ('hg_s2c16', 2, 16, 32), ('hg_s1c128', 1, 128, 33)])
def test_hourglass_train_step(tag, stacks, chan, seed):
    g = load(tag + '.npz')
    torch.set_num_threads(8)
    net = om.create_hg(stacks, 1, 16, chan)
    assert sum(p.numel() for p in net.parameters()) == int(g['nparams'])
    om.deterministic_fill_(net, seed=seed)
    net.train()
    img = t(inputs.images(seed + 100, 2, 128))
    heat = t(inputs.heatmaps_from_pts(inputs.heat_pts(seed + 200, 2, res=32), res=32))
    opt = ostep.make_optimizer(net)
    out, loss = ostep.pose_loss_and_grads(net, img, heat)
    grads = [p.grad.clone() for p in net.parameters()]
    opt.step()
    assert np.allclose(np.stack([o.detach().numpy() for o in out]), g['out'], atol=2e-5)
    assert abs(float(loss) - float(g['loss'])) < 1e-6
    assert np.allclose(digest(grads), g['grad_digest'], rtol=1e-3, atol=1e-6)
    if 'grads' in g:
        assert np.allclose(np.concatenate([x.flatten().numpy() for x in grads]), g['grads'], rtol=1e-3, atol=1e-7)
    if 'head_grads' in g:
        names = [n for n, _ in net.named_parameters()]
        assert [n for n in names if n.startswith('out_conv.0.') or n.startswith('linear.0.1.')] == list(g['head_names'])
        mine = np.concatenate([x.flatten().numpy() for n, x in zip(names, grads) if n.startswith('out_conv.0.') or n.startswith('linear.0.1.')])
        assert np.allclose(mine, g['head_grads'], rtol=1e-3, atol=1e-7)
    # RMSprop's first step is ~ +-10 lr * sign(g): near-zero gradients (biases in front of a
    # BN) are rounding noise whose sign is not reproducible, so compare digests loosely ...
    assert np.allclose(digest([p.detach() for p in net.parameters()])[:, 1], g['param_digest'][:, 1], rtol=2e-2, atol=1e-3)
    assert np.allclose(digest([b.float() for b in net.buffers()]), g['buf_digest'], rtol=1e-4, atol=1e-5)
    acc = opl.accuracy(out[-1].detach(), heat, [0, 1, 2, 3, 4, 5, 10, 11, 14, 15])
    assert np.allclose(acc.numpy(), g['acc'], atol=1e-6)
    net.eval()
    with torch.no_grad():
        oe = np.stack([o.numpy() for o in net(img)])
    assert np.allclose(oe, g['out_eval'], rtol=1e-3, atol=2e-3)


def test_rmsprop_update_matches_torch():
    p = t(inputs.rng(51).standard_normal(1000).astype(np.float32))
    gr = t(inputs.rng(52).standard_normal(1000).astype(np.float32))
    q = torch.nn.Parameter(p.clone())
    opt = torch.optim.RMSprop([q], lr=2.5e-4, alpha=0.99, eps=1e-8)
    v = torch.zeros_like(p)
    for _ in range(3):
        q.grad = gr.clone()
        opt.step()
        ostep.rmsprop_update(p, gr, v, 2.5e-4)
    assert torch.equal(p, q.detach())


def test_agent_half_hourglass():
    g = load('asn_c16.npz')
    net = om.create_hg(2, 1, 16, 16)
    asn = om.create_asn(16, 16, 7, 7, is_aug=True)
    assert sum(p.numel() for p in asn.parameters()) == int(g['nparams'])
    om.deterministic_fill_(net, seed=41)
    om.deterministic_fill_(asn, seed=42)
    net.eval(); asn.train()
    img = t(inputs.images(141, 2, 256))
    ls, lr = ostep.agent_logits(net, asn, img)
    assert np.allclose(ls.detach().numpy(), g['logits_s'], atol=1e-5)
    assert np.allclose(lr.detach().numpy(), g['logits_r'], atol=1e-5)
    ps, pr = torch.softmax(ls, 1), torch.softmax(lr, 1)
    gs = opl.gen_groundtruth(ps, t(g['idx_s']), t(g['reg']), t(g['ag']))
    gr = opl.gen_groundtruth(pr, t(g['idx_r']), t(g['ag']), t(g['reg']))
    assert np.allclose(gs.numpy(), g['gs'], atol=1e-6) and np.allclose(gr.numpy(), g['gr'], atol=1e-6)
    loss = ostep.agent_kl_loss(ls, lr, gs, gr)
    assert abs(float(loss) - float(g['loss'])) < 1e-6
    net.zero_grad(); asn.zero_grad()
    loss.backward()
    assert all(p.grad is None for p in net.parameters())
    assert np.allclose(digest([p.grad for p in asn.parameters()]), g['grad_digest'], rtol=1e-3, atol=1e-7)


def test_census_full_size():
    g = load('census.npz')
    net = om.create_hg(2, 1, 16, 256)
    assert sum(p.numel() for p in net.parameters()) == int(g['n_hg']) == 6570784
    assert list(net.state_dict().keys()) == [str(k) for k in g['hg_keys']]
    asn = om.create_asn(256, 256, 7, 7, is_aug=True)
    assert sum(p.numel() for p in asn.parameters()) == int(g['n_asn']) == 2577934


def test_occlusion_agent_branch():
    """mask logits, the reference's own draw (np.random.seed replay), the masked two-stack
    forward, the pose-net gradients through the masks and the agent's gradients -- against the transliterated reference."""
    import copy
    g = load('dropout_c16.npz')
    net = om.create_hg(2, 1, 16, 16)
    asn = om.create_asn(16, 16, is_dropout=True)
    assert list(asn.state_dict().keys()) == [str(k) for k in g['asn_keys']]
    assert sum(p.numel() for p in asn.parameters()) == int(g['nparams'])
    om.deterministic_fill_(net, seed=51)
    om.deterministic_fill_(asn, seed=52)
    net.train(); asn.train()
    img = t(inputs.images(151, 2, 256))
    heat = t(inputs.heatmaps_from_pts(inputs.heat_pts(152, 2, res=64), res=64))
    half = copy.deepcopy(net)(img, copy.deepcopy(asn), is_half_hg=True, is_dropout=True)
    assert np.allclose(half.detach().numpy(), g['pred_mask'], atol=1e-5)
    np.random.seed(153)
    out, pred_mask, indexes = net(img, asn, is_dropout=True)
    assert np.array_equal(indexes.numpy(), g['indexes'])
    assert np.allclose(pred_mask.detach().numpy(), g['pred_mask'], atol=1e-5)
    assert np.allclose(np.stack([o.detach().numpy() for o in out]), g['out'], atol=2e-5)
    loss = sum(((o - heat) ** 2).sum() / o.numel() for o in out)
    assert abs(float(loss) - float(g['loss'])) < 1e-6
    net.zero_grad(); asn.zero_grad()
    loss.backward(retain_graph=True)
    assert all(p.grad is None for p in asn.parameters())
    assert np.allclose(digest([p.grad for p in net.parameters()]), g['pose_grad_digest'], rtol=2e-3, atol=1e-6)
    pred_mask.backward(t(inputs.rng(154).standard_normal((2, 1, 4, 4)).astype(np.float32)))
    assert np.allclose(digest([p.grad for p in asn.parameters()]), g['asn_grad_digest'], rtol=2e-3, atol=1e-6)
    # _dropout and _sample_mask on their own
    x = t(inputs.rng(155).standard_normal((2, 3, 16, 16)).astype(np.float32))
    assert np.array_equal(om.Hourglass.dropout(x, t(g['drop_masks'])).numpy(), g['drop_x_out'])
    np.random.seed(156)
    lg = t(inputs.rng(157).normal(0, 2.0, (6, 1, 4, 4)).astype(np.float32))
    smask, sidx = om.sample_mask(lg)
    assert np.array_equal(sidx.numpy(), g['sample_idx']) and np.array_equal(smask.numpy(), g['sample_masks'])
    assert np.array_equal(om.masks_from_indexes(sidx).numpy(), g['sample_masks'])


def test_occlusion_sampler_law():
    """sequential inverse-CDF draws == the distribution of np.random.choice(K, 2, p, replace=False)"""
    rng = np.random.RandomState(5)
    p = np.array([0.5, 0.25, 0.15, 0.1])
    n = 40000
    ref = np.stack([rng.choice(4, 2, p=p, replace=False) for _ in range(n)])
    mine = om.sample_cells_inverse_cdf(np.tile(p, (n, 1)), np.random.RandomState(6).random_sample((n, 2)))
    assert (mine[:, 0] != mine[:, 1]).all()
    for a in range(4):
        for b in range(4):
            if a == b:
                continue
            want = p[a] * p[b] / (1 - p[a])
            f_ref = np.mean((ref[:, 0] == a) & (ref[:, 1] == b))
            f_mine = np.mean((mine[:, 0] == a) & (mine[:, 1] == b))
            assert abs(f_ref - want) < 0.01 and abs(f_mine - want) < 0.01, (a, b, want, f_ref, f_mine)


# ------------------------------------------------------------------ crop (row a9)
def test_crop_restatement_reproduces_the_reference_crop():
    """oracle.crop (numpy restatement of scipy.misc.pilutil + Pillow's resize / rotate arithmetic) == the reference's own
    crop() over the real Pillow (tests/golden/crop.npz), byte for byte, for all 15 cases; quirk=False (the device
    specification) equals it wherever the crop holds a black and a white pixel."""
    from oracle import crop as oc
    g = load('crop.npz')
    frames = {}
    for i, (kind, c0, s0, r0, flip, gain, neutral) in enumerate(inputs.WARP_CASES):
        if kind not in frames:
            frames[kind] = inputs.warp_frame(kind)
        c = np.array(c0, dtype=np.float32)
        if flip:
            c[0] = np.float32(1280) - c[0]
        for quirk in ((True, False) if neutral else (True,)):
            out = oc.crop(oc.source_image(frames[kind], flip, gain), c, np.float32(s0), r0, 256, 200, quirk=quirk)
            assert out.dtype == np.uint8 and out.shape == (256, 256, 3)
            assert np.array_equal(out[1::4, 2::4], g['crop%02d_sub' % i]), (i, quirk)
            sums = [int(out[..., k].astype(np.int64).sum()) for k in range(3)] + [int((out[..., k].astype(np.int64) ** 2).sum()) for k in range(3)]
            assert sums == [int(v) for v in g['crop%02d_sums' % i]], (i, quirk)
            if 'crop%02d_full' % i in g.files:
                assert np.array_equal(out, g['crop%02d_full' % i])
    assert [int(frames[k].astype(np.int64).sum()) for k in sorted(frames)] == [int(v) for v in g['frame_sums'].reshape(-1)]


def test_pillow_restatements_equal_pillow():
    """the two PIL restatements against the real library on random images (skipped where Pillow is absent)"""
    Image = pytest.importorskip('PIL.Image')
    from oracle import crop as oc
    rng = inputs.rng(97)
    for h, w, ang in [(37, 53, 25.0), (64, 64, -40.0), (101, 77, 58.0), (50, 50, 90.0), (50, 60, 90.0), (40, 40, 180.0), (33, 47, -17.5), (64, 64, 270.0)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(oc.pil_rotate_bilinear(a, ang), np.array(Image.fromarray(a).rotate(ang, resample=Image.BILINEAR)))
    for h, w, oh, ow in [(37, 53, 20, 30), (360, 640, 112, 200), (402, 402, 256, 256), (180, 180, 256, 256), (511, 509, 256, 256), (300, 256, 256, 256), (256, 300, 256, 256)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(oc.pil_resize_bilinear(a, ow, oh), np.array(Image.fromarray(a).resize((ow, oh), resample=Image.BILINEAR)))


def test_augmentation_law_restatements():
    """oracle.regular_aug / agent_aug: clips, the 0.6 / 0.5 thresholds (<=), fp32 scale product, the mirror of the centre"""
    c, s, r, flip, gains = opl.regular_aug([640.25, 360.5], 2.5, 1280.0, 3.0, -3.0, 0.7, 0.5, [0.0, 0.5, 1.0])
    assert s == float(np.float32(2.5) * np.float32(2 ** 0.5)) and r == -60.0 and flip and c[0] == 1280 - 640.25
    assert np.allclose(gains, [0.6, 1.0, 1.4])
    assert opl.regular_aug([1, 1], 1.0, 10.0, 0.0, 1.0, 0.6, 0.51, [0, 0, 0])[2:4] == (0.0, False)
    c, s, r, flip, _ = opl.agent_aug([100.0, 50.0], 2.0, 1280.0, 0, 6, -5.0, 5.0, 0.9, [0, 0, 0])
    assert abs(np.log2(s / 2.0) - (-0.6 - 0.05 + 1e-3)) < 1e-6 and r == 65.0 and not flip


def _agent_draws(seed):
    """np.random call order of AGENT.__getitem__ with bins given: randn (scale), randn (rotation), random (flip), 3 gains"""
    st = np.random.RandomState(seed)
    return np.array([st.randn(), st.randn()] + [st.random_sample() for _ in range(4)], dtype=np.float64)


def _check_sample(D, tag, got):
    inp, heat, c, s, r, pts, normalizer = got
    b = np.rint(inp * 255.0).astype(np.uint8)
    assert np.array_equal(b[:, 1::4, 2::4], D[tag + '_inp_sub']), tag
    sums = np.array([b[k].astype(np.int64).sum() for k in range(3)] + [(b[k].astype(np.int64) ** 2).sum() for k in range(3)])
    assert np.array_equal(sums, D[tag + '_inp_sums']), tag
    assert np.array_equal(heat, D[tag + '_heat']), tag                       # bit-exact heat maps
    assert np.array_equal(np.asarray(c, np.float32), D[tag + '_c']) and np.array_equal(np.asarray(s, np.float32).reshape(-1), D[tag + '_s'].reshape(-1)), tag
    assert np.array_equal(np.asarray(r, np.float32).reshape(-1), D[tag + '_r'].reshape(-1)), tag
    assert np.array_equal(np.asarray(pts, np.float32), D[tag + '_pts']) and float(normalizer) == float(D[tag + '_normalizer']), tag


def test_whole_dataset_samples_equal_the_reference():
    """tests/golden/dataset.npz = the reference's MPII.__getitem__ (data/mpii_for_mpii.py:83-163; train with np.random seeded,
    and val) and AGENT.__getitem__ (data/joint_train_s_r_agent.py:98-177; bins given, with and without separate_s_r) on the
    3-person JSON + PNG set of tests/inputs.py: the oracle's composition reproduces every sample -- crop bytes, heat maps,
    c, s, r, joints, normaliser -- exactly."""
    D = load('dataset.npz')
    frames, anno = inputs.dataset_people()
    assert np.array_equal(np.array([int(f.astype(np.int64).sum()) for f in frames]), D['frame_sums'])
    train = [0, 1]
    for index in (0, 1):
        for seed in (11, 12, 13, 14, 15, 16):
            got = opl.mpii_getitem(frames[train[index]], anno[train[index]], inputs.legacy_draws(seed), is_train=True)
            _check_sample(D, 'train%d_seed%d' % (index, seed), got)
    _check_sample(D, 'val0', opl.mpii_getitem(frames[2], anno[2], None, is_train=False))
    full = np.rint(opl.mpii_getitem(frames[0], anno[0], inputs.legacy_draws(11), is_train=True)[0] * 255).astype(np.uint8)
    assert np.array_equal(full, D['train0_seed11_inp_full'])
    for k in (0, 1):
        i = int(D['agent_img_index'][k])
        got = opl.agent_getitem(frames[train[i]], anno[train[i]], int(D['agent_scale_index'][k]), int(D['agent_rot_index'][k]), _agent_draws(21 + k))
        _check_sample(D, 'agent%d' % k, got)
    sep = opl.agent_getitem(frames[0], anno[0], 6, 2, _agent_draws(31), separate_s_r=True)
    _check_sample(D, 'sep_scale', sep[0])
    _check_sample(D, 'sep_rot', sep[1])
