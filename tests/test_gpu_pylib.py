"""GPU parity of the pose-library kernels (through the C ABI / the pylib facade) against the golden
vectors produced by the reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests import inputs
from tests.test_oracle_golden import load, t, _eval_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def P():
    from pose_adv_aug_amd import pylib
    return pylib


def test_heatmaps_bit_exact(P):
    g = load('pylib.npz')
    hm, valid = P.HumanPts.pts2heatmap(g['hm_special_pts'].copy(), [64, 64])
    assert np.array_equal(hm.astype(np.float32), g['hm_special'])
    assert np.array_equal(valid, g['hm_special_valid'])
    pts = inputs.heat_pts(11, 4)
    hms = P.HumanPts.pts2heatmap_batch(pts, 64, 64).cpu().numpy()
    assert np.array_equal(hms[:2], g['hm_rand'])
    dg = np.array([[h.astype(np.float64).sum(), (h.astype(np.float64) ** 2).sum(), h.max()] for h in hms.reshape(-1, 64, 64)])
    assert np.allclose(dg, g['hm_rand_digest'], rtol=1e-6, atol=1e-6)
    # full-size property: 24 x 16 maps, each map is either empty or has its peak == 1 at int(pt) (trunc toward zero)
    big = inputs.heat_pts(5, 24)
    out = P.HumanPts.pts2heatmap_batch(big, 64, 64)
    from oracle import pylib as opl
    ref = np.stack([opl.pts2heatmap(big[i].copy(), [64, 64])[0] for i in range(24)]).astype(np.float32)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_transforms(P):
    g = load('pylib.npz')
    c, s, r, pts = g['tf_c'], g['tf_s'], g['tf_r'], g['tf_pts']
    from oracle import pylib as opl
    for i in range(6):
        # float64 on the device (sin/cos may differ from numpy's in the last ulp)
        assert np.allclose(P.HumanAug.GetTransform(c[i], s[i], r[i], 256, 200), g['tf_T256'][i], rtol=1e-13, atol=1e-11)
        assert np.allclose(P.HumanAug.GetTransform(c[i], s[i], r[i], 64, 200), g['tf_T64'][i], rtol=1e-13, atol=1e-11)
        p64 = P.HumanAug.TransformPts(pts[i], c[i], s[i], r[i], 64, 200)
        assert np.allclose(p64, g['tf_pts64'][i], rtol=1e-12, atol=1e-10)
        back = P.Evaluation.TransformPts(g['tf_pts64'][i] + 1, c[i], s[i], r[i], 64, 200, invert=1)
        assert np.array_equal(back, g['tf_pts64_eval_inv'][i])
        sh = P.HumanAug.shufflelr(t(pts[i].copy()), width=1280).numpy()
        assert np.array_equal(sh, g['tf_shufflelr'][i])
    # batched device path == per-sample path, incl. flip + joint swap + invalid joints
    params = P.HumanAug.make_params(c, s, r, flip=[0, 1, 0, 1, 1, 0])
    params[:, 0] = torch.where(params[:, 4] > 0, 1280 - params[:, 0], params[:, 0])
    T, Tinv = P.HumanAug.affine_params(params, 256, 64)
    out, img_pts = P.HumanAug.transform_pts_batch(pts, params, T, 1280)
    from oracle import pylib as opl
    for i in range(6):
        flip = i in (1, 3, 4)
        # the reference mirrors the joints as a float32 torch tensor (data/mpii_for_mpii.py:128-129)
        p = opl.shufflelr(pts[i].astype(np.float32), np.float32(1280)) if flip else pts[i].astype(np.float32)
        cc = c[i].copy()
        if flip:
            cc[0] = 1280 - cc[0]
        ref = opl.transform_pts(p.astype(np.float64), cc, s[i], r[i], 64)
        bad = (p[:, 0] <= 0) | (p[:, 1] <= 0)
        ref[bad] = 0
        assert np.allclose(out[i].cpu().numpy(), ref, rtol=1e-9, atol=1e-6)
        assert np.allclose(img_pts[i].cpu().numpy(), p.astype(np.float32), atol=1e-3)


def test_evaluation_and_humanacc(P):
    g = load('pylib.npz')
    c, s, r, gpts, norm, tgt, pred = _eval_inputs(g)
    n = 6
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]
    E = P.Evaluation
    assert np.array_equal(E.get_preds(t(pred)).cpu().numpy(), g['ev_get_preds'])
    assert np.allclose(E.accuracy(t(pred), t(tgt), idx).cpu().numpy(), g['ev_accuracy'], atol=1e-4)
    cT, sT, rT = t(c).float(), t(s).float().view(n, 1), t(r).float().view(n, 1)
    assert np.array_equal(E.final_preds(t(pred), cT, sT, [64, 64], rT).cpu().numpy(), g['ev_final_preds'])
    a = E.accuracy_origin_res(t(pred), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)
    assert np.allclose(a.cpu().numpy(), g['ev_acc_origin'], atol=1e-4)
    pp = E.per_person_pckh(t(pred), t(tgt), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)
    assert np.allclose(pp.cpu().numpy(), g['ev_per_person'], atol=1e-4)
    pk = P.HumanAcc.approx_PCKh(E.get_preds(t(pred)), E.get_preds(t(tgt)), idx, 64)
    assert abs(pk - float(g['acc_approx_pckh'])) < 1e-4
    fl = P.HumanAug.shuffle_channels_for_horizontal_flipping(P.HumanAug.flip_channels(t(pred[:1].copy())))
    assert np.array_equal(fl.numpy(), g['flip_maps'])
    # the rest of the HumanAcc family (pylib/HumanAcc.py:46-308) against the reference's goldens
    A = P.HumanAcc
    pp_, gp_ = E.get_preds(t(pred)), E.get_preds(t(tgt))
    avg, per = A.approx_PCKh_per(pp_, gp_, idx, 64)
    assert abs(avg - float(g['acc_per_avg'])) < 1e-4 and np.allclose(per.cpu().numpy(), g['acc_per'], atol=1e-4)
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert A.PCKh(pp_, gp_, t(norm).float() / 20) is None
    printed = np.array([float(l.split(':')[1]) for l in buf.getvalue().strip().splitlines()])
    assert np.allclose(printed, g['acc_pckh_print'], atol=1.01e-4)
    assert np.array_equal(A.approx_PCKh_samples(pp_, gp_, 64).cpu().numpy(), g['acc_samples'])
    assert np.array_equal(A.correct_predicted_joints(pp_, gp_, 64).cpu().numpy(), g['acc_correct'])
    assert np.array_equal(A.correct_predicted_joints_original_resolution(pp_, gp_, 2.5).cpu().numpy(), g['acc_correct_orig'])
    assert np.allclose(A.predicted_joints_dist_to_grnd(pp_, gp_, 64).cpu().numpy(), g['acc_dist_to_grnd'], atol=1e-6)
    # calc_dists matrix vs the oracle
    from oracle import pylib as opl
    d = E.calc_dists(E.get_preds(t(pred)), E.get_preds(t(tgt)), torch.ones(n) * 6.4).cpu()
    dref = opl.calc_dists(opl.get_preds(t(pred)), opl.get_preds(t(tgt)), torch.ones(n) * 6.4)
    assert torch.allclose(d, dref, atol=1e-6)


def test_pckh_full_batch_against_oracle(P):
    """BASELINE config size (B=24): PCKh of noisy maps, device vs CPU oracle, must agree to 1e-4."""
    from oracle import pylib as opl
    c, s, r, gpts, norm = inputs.person_meta(77, 24)
    tp = np.stack([opl.transform_pts(gpts[i], c[i], s[i], r[i], 64) for i in range(24)])
    tp[gpts[..., 0] <= 0] = 0
    tgt = inputs.heatmaps_from_pts(tp, 64)
    pred = inputs.noisy_heatmaps(78, tgt, noise=0.25)
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]
    E = P.Evaluation
    cT, sT, rT = t(c).float(), t(s).float().view(24, 1), t(r).float().view(24, 1)
    a_dev = E.accuracy(t(pred), t(tgt), idx).cpu().numpy()
    a_ref = opl.accuracy(t(pred), t(tgt), idx).numpy()
    assert np.allclose(a_dev, a_ref, atol=1e-4)
    o_dev = E.accuracy_origin_res(t(pred), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).cpu().numpy()
    o_ref = opl.accuracy_origin_res(t(pred), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).numpy()
    assert np.allclose(o_dev, o_ref, atol=1e-4)
    p_dev = E.per_person_pckh(t(pred), t(tgt), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).cpu().numpy()
    p_ref = opl.per_person_pckh(t(pred), t(tgt), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).numpy()
    assert np.allclose(p_dev, p_ref, atol=1e-4)
    k_dev = P.HumanAcc.approx_PCKh(E.get_preds(t(pred)), E.get_preds(t(tgt)), idx, 64)
    k_ref = opl.approx_pckh(opl.get_preds(t(pred)), opl.get_preds(t(tgt)), idx, 64)
    assert abs(k_dev - k_ref) < 1e-4


def test_losses(P):
    g = load('pylib.npz')
    _, _, _, _, _, tgt, pred = _eval_inputs(g)
    w = inputs.rng(14).random(pred.shape, dtype=np.float32) + 0.5
    assert abs(float(P.Criterion.weighted_L2(t(pred), t(tgt), t(w))) - float(g['l2_weighted'])) < 1e-6
    assert abs(float(P.Criterion.weighted_L2(t(pred), t(tgt), 1)) - float(g['l2_unit'])) < 1e-6


def test_warp_matches_oracle_sampler(P):
    """the pure inverse-affine sampler (an operator; the loops' crop is pa_crop: tests/test_gpu_crop.py)"""
    from oracle import pylib as opl
    rng = inputs.rng(90)
    frames = rng.integers(0, 256, size=(3, 90, 120, 3), dtype=np.uint8)
    c = np.array([[60.0, 45.0], [50.0, 40.0], [70.0, 30.0]])
    s = np.array([0.35, 0.5, 0.28])
    r = np.array([0.0, 25.0, -40.0])
    flip = [0, 1, 0]
    gain = np.array([[1, 1, 1], [0.7, 1.3, 1.0], [1.4, 0.6, 1.2]], dtype=np.float32)
    params = P.HumanAug.make_params(c, s, r, flip=flip, gain=gain)
    _, tinv = P.HumanAug.affine_params(params, 64, 16)
    out4, outf = P.HumanAug.warp_batch(frames, tinv, params, res=64, want_nchw=True)
    for i in range(3):
        ref = opl.warp_bilinear(frames[i], c[i], s[i], r[i], 64, flip=bool(flip[i]), gain=gain[i])
        assert np.allclose(outf[i].cpu().numpy(), ref, atol=2e-6)
        nhwc = out4[i].float().cpu().numpy()
        assert np.allclose(nhwc[..., :3].transpose(2, 0, 1), ref, atol=4e-3) and np.all(nhwc[..., 3] == 0)


def test_rmsprop_matches_torch():
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    n = 100003
    p0 = t(inputs.rng(51).standard_normal(n).astype(np.float32))
    g0 = t(inputs.rng(52).standard_normal(n).astype(np.float32))
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.RMSprop([q], lr=2.5e-4, alpha=0.99, eps=1e-8)
    p, g, v = p0.cuda(), g0.cuda(), torch.zeros(n, device='cuda')
    for _ in range(3):
        q.grad = g0.clone()
        opt.step()
        check(lib().pa_rmsprop_step(ptr(p), ptr(g), ptr(v), n, 2.5e-4, 0.99, 1e-8, 1.0, stream()))
    assert torch.allclose(p.cpu(), q.detach(), rtol=1e-6, atol=1e-7)


def test_samplers_follow_the_reference_laws():
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    B = 4096
    meta = torch.tensor([[640.0, 360.0, 2.5, 1280.0]], device='cuda').repeat(B, 1).contiguous()
    params = torch.zeros(B, 8, device='cuda', dtype=torch.float64)
    check(lib().pa_sample_aug(ptr(meta), None, None, 0, 7, 3, B, ptr(params), stream()))
    p = params.cpu().numpy()
    sc = np.log2(p[:, 2] / 2.5)
    assert sc.min() >= -0.5 - 1e-5 and sc.max() <= 0.5 + 1e-5 and 0.15 < sc.std() < 0.3
    assert np.abs(p[:, 3]).max() <= 60 + 1e-4 and 0.55 < (p[:, 3] == 0).mean() < 0.68
    assert 0.45 < p[:, 4].mean() < 0.55
    assert np.all((p[:, 0] == 640.0)) and p[:, 5:].min() >= 0.6 and p[:, 5:].max() <= 1.4
    # deterministic in (seed, step)
    params2 = torch.zeros(B, 8, device='cuda', dtype=torch.float64)
    check(lib().pa_sample_aug(ptr(meta), None, None, 0, 7, 3, B, ptr(params2), stream()))
    assert torch.equal(params, params2)
    # agent law: bins
    si = torch.randint(0, 7, (B,), device='cuda', dtype=torch.int32)
    ri = torch.randint(0, 7, (B,), device='cuda', dtype=torch.int32)
    check(lib().pa_sample_aug(ptr(meta), ptr(si), ptr(ri), 1, 7, 4, B, ptr(params), stream()))
    p = params.cpu().numpy()
    mu_s = -0.6 + 0.2 * si.cpu().numpy(); mu_r = -60 + 20.0 * ri.cpu().numpy()
    f = np.log2(p[:, 2] / 2.5)
    assert np.all(f >= mu_s - 0.05 + 1e-3 - 1e-5) and np.all(f <= mu_s + 0.05 + 1e-5)
    assert np.all(p[:, 3] >= mu_r - 5 + 1e-3 - 1e-4) and np.all(p[:, 3] <= mu_r + 5 + 1e-4)
    # categorical sampling frequencies
    logits = torch.tensor([[0.0, 1.0, 2.0, -1.0, 0.5, 0.0, 0.3]], device='cuda').repeat(B, 1).contiguous()
    probs = torch.zeros(B, 7, device='cuda'); idx = torch.zeros(B, dtype=torch.int32, device='cuda')
    check(lib().pa_sample_categorical(ptr(logits), B, 7, 11, 0, 0, ptr(probs), ptr(idx), stream()))
    ref = torch.softmax(logits[0].cpu(), 0).numpy()
    assert np.allclose(probs[0].cpu().numpy(), ref, atol=1e-6)
    freq = np.bincount(idx.cpu().numpy(), minlength=7) / B
    assert np.abs(freq - ref).max() < 0.03


def test_metric_edge_cases_against_oracle(P):
    """Degenerate inputs of the metrics (pylib/Evaluation.py:21-22,41-52,60-75,105-167), device vs oracle: a person with NO
    annotated joint, a joint annotated on nobody (its per-joint accuracy is -1 and is left out of the mean), all-negative and
    all-zero heat maps (prediction (0, 0)), a peak on the first row / column (ground truth there counts as invalid in heat-map
    mode), ties (first maximum wins), a single sample, thresholds hit exactly."""
    from oracle import pylib as opl
    E = P.Evaluation
    B = 5
    c, s, r, gpts, norm = inputs.person_meta(91, B)
    gpts[1] = 0.0                                   # nobody annotated on sample 1
    gpts[:, 7] = 0.0                                # joint 7 annotated on no sample
    tp = np.stack([opl.transform_pts(gpts[i], c[i], s[i], r[i], 64) for i in range(B)])
    tp[gpts[..., 0] <= 0] = 0
    tp[2, 3] = [0.5, 20.0]                          # Gaussian peak in column 0 -> 1-based x = 1: not > 1
    tgt = inputs.heatmaps_from_pts(tp, 64)
    pred = inputs.noisy_heatmaps(92, tgt, noise=0.1)
    pred[0, 2] = -1.0                               # all negative
    pred[0, 4] = 0.0                                # all zero
    pred[3, 5] = 0.25                               # a plateau: every pixel ties
    pred[4, 6, 10, 20] = pred[4, 6, 40, 50] = 9.0   # two equal maxima
    idx = list(range(16))
    cT, sT, rT = t(c).float(), t(s).float().view(B, 1), t(r).float().view(B, 1)
    for sl in (slice(0, B), slice(1, 2)):           # the whole batch, and the unannotated person alone
        pr, tg = t(pred[sl]), t(tgt[sl])
        args = (cT[sl], sT[sl], [64, 64], t(gpts[sl]).float(), t(norm[sl]).float(), rT[sl])
        assert np.array_equal(E.get_preds(pr).cpu().numpy(), opl.get_preds(pr).numpy())
        assert np.allclose(E.accuracy(pr, tg, idx).cpu().numpy(), opl.accuracy(pr, tg, idx).numpy(), atol=1e-6)
        assert np.array_equal(E.final_preds(pr, cT[sl], sT[sl], [64, 64], rT[sl]).cpu().numpy(), opl.final_preds(pr, cT[sl], sT[sl], [64, 64], rT[sl]).numpy())
        assert np.allclose(E.accuracy_origin_res(pr, *args).cpu().numpy(), opl.accuracy_origin_res(pr, *args).numpy(), atol=1e-6)
        assert np.allclose(E.per_person_pckh(pr, tg, *args).cpu().numpy(), opl.per_person_pckh(pr, tg, *args).numpy(), atol=1e-6)
    a = opl.accuracy(t(pred), t(tgt), idx).numpy()
    assert a[8] == -1 and float(opl.per_person_pckh(t(pred), t(tgt), cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)[1]) == 0.0
    # heat maps and the loss at the borders of the validity rule (pylib/HumanPts.py:41-43): x or y <= 0, > W, == W
    pts = np.zeros((1, 16, 2)); pts[0, :6] = [[64.0, 64.0], [64.001, 10.0], [1e-9, 1e-9], [0.0, 5.0], [63.999, 0.001], [32.5, 64.0]]
    hm = P.HumanPts.pts2heatmap_batch(torch.from_numpy(pts).cuda(), 64, 64).cpu().numpy()
    assert np.array_equal(hm[0], inputs.heatmaps_from_pts(pts, 64)[0])


def test_params_csr_and_the_copy_probe_forms():
    """pa_params_csr: the fp32 c / s / r the metric calls take = the float64 parameter block rounded once (what `.float()` of its columns gave
    through three framework kernels until round 6); pa_copy_probe_form: every form of the calibration copy moves the bytes."""
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    g = inputs.rng(77)
    B = 24
    params = torch.from_numpy(g.uniform(-300, 1300, size=(B, 8))).cuda()
    csr = torch.full((4 * B,), float('nan'), device='cuda')
    check(lib().pa_params_csr(ptr(params), B, ptr(csr), stream()), 'pa_params_csr')
    assert torch.equal(csr[:2 * B].view(B, 2), params[:, 0:2].float())
    assert torch.equal(csr[2 * B:3 * B], params[:, 2].float()) and torch.equal(csr[3 * B:], params[:, 3].float())
    src = torch.randint(0, 255, (3 * 1024 * 1024 + 16,), dtype=torch.uint8, device='cuda')
    for form in (0, 1, 2):
        dst = torch.zeros_like(src)
        check(lib().pa_copy_probe_form(ptr(dst), ptr(src), src.numel(), form, stream()), 'pa_copy_probe_form')
        assert torch.equal(dst, src), form
