#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF in the build
container.  Run from the repo root:  python tests/golden/make_goldens.py

The reference (/root/reference) is Python-2 / torch-0.3 source.  This script
makes a throw-away transliteration under /tmp (never in the repo, never
shipped), imports it, feeds it the seeded inputs of tests/inputs.py and the
deterministic weights of oracle.model.deterministic_fill_, and stores inputs'
seeds + the reference's outputs.  Steps:
  1. copy the checkout to /tmp/ref3, drop stale *.pyc
  2. lib2to3 fixers print/import/dict/xrange
  3. restore py2 integer division where it matters (out_num/2, height / 4, res/10)
  4. numpy-2 strictness: squeeze 1-element scale/rot arrays handed to GetTransform
  5. occlusion branch only: drop the `.cuda()` of _sample_mask's mask buffer (no GPU here), `/ width` -> `// width`
Only data (arrays) is written to the repo.
"""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
REF = '/root/reference'
TMP = '/tmp/ref3_goldens'
OUT = os.path.join(REPO, 'tests', 'golden')


def transliterate():
    if os.path.isdir(TMP):
        shutil.rmtree(TMP)
    shutil.copytree(REF, TMP)
    for root, _, files in os.walk(TMP):
        os.chmod(root, 0o755)
        for f in files:
            p = os.path.join(root, f)
            os.chmod(p, 0o644)
            if f.endswith('.pyc'):
                os.remove(p)
    subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n', '-f', 'print', '-f', 'import',
                    '-f', 'dict', '-f', 'xrange', TMP], check=True, capture_output=True)

    def patch(rel, subs):
        p = os.path.join(TMP, rel)
        s = open(p).read()
        for a, b in subs:
            s, n = re.subn(a, b, s)
            assert n > 0, (rel, a)
        open(p, 'w').write(s)
    patch('models/asn_stacked_hg.py', [(r'out_num/2', 'out_num//2'), (r'height / 4', 'height // 4'),
                                       # occlusion branch (:102-136): no GPU in this container; py2 integer division
                                       (r'torch\.ones\(pred_masks\.size\(\)\)\.cuda\(\)', 'torch.ones(pred_masks.size())'),
                                       (r'dropout_indexes\[j\] / width', 'dropout_indexes[j] // width')])
    patch('pylib/HumanAcc.py', [(r'normalize = res/10', 'normalize = res//10')])
    sys.path.insert(0, TMP)


def load_ref():
    transliterate()
    install_scipy_misc_shim()
    from pylib import HumanPts, HumanAug, Evaluation, HumanAcc, Criterion
    from models import asn_stacked_hg
    from utils import util

    def squeeze_args(fn):
        def w(center, scale, rot, res, size):
            return fn(center, float(np.asarray(scale).reshape(-1)[0]), float(np.asarray(rot).reshape(-1)[0]), res, size)
        return w
    HumanAug.GetTransform = squeeze_args(HumanAug.GetTransform)
    Evaluation.GetTransform = squeeze_args(Evaluation.GetTransform)
    return dict(HumanPts=HumanPts, HumanAug=HumanAug, Evaluation=Evaluation, HumanAcc=HumanAcc,
                Criterion=Criterion, M=asn_stacked_hg, util=util)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def install_scipy_misc_shim():
    """scipy.misc.{bytescale, toimage, fromimage, imresize, imrotate} were removed in scipy 1.3 (this image has 1.15); the
    reference's crop calls them (pylib/HumanAug.py:130,169,175).  Restated here from scipy 0.19..1.2 `misc/pilutil.py`
    (3-D RGB arrays only) on top of the REAL Pillow of this container, which does all the pixel work."""
    import scipy.misc
    from PIL import Image

    def bytescale(data, cmin=None, cmax=None, high=255, low=0):
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

    def toimage(arr, high=255, low=0, cmin=None, cmax=None, pal=None, mode=None, channel_axis=None):
        data = np.asarray(arr)
        assert data.ndim == 3 and data.shape[2] == 3 and data.shape[0] != 3 and data.shape[1] != 3
        bytedata = bytescale(data, high=high, low=low, cmin=cmin, cmax=cmax)
        return Image.frombytes('RGB', (data.shape[1], data.shape[0]), bytedata.tobytes())

    def fromimage(im):
        return np.array(im)

    func = {'nearest': 0, 'lanczos': 1, 'bilinear': 2, 'bicubic': 3, 'cubic': 3}

    def imresize(arr, size, interp='bilinear', mode=None):
        im = toimage(arr, mode=mode)
        ts = type(size)
        if np.issubdtype(ts, np.signedinteger):
            size = tuple((np.array(im.size) * (size / 100.0)).astype(int))
        elif np.issubdtype(ts, np.floating):
            size = tuple((np.array(im.size) * size).astype(int))
        else:
            size = (size[1], size[0])
        return fromimage(im.resize(tuple(int(v) for v in size), resample=func[interp]))

    def imrotate(arr, angle, interp='bilinear'):
        im = toimage(np.asarray(arr))
        return fromimage(im.rotate(angle, resample=func[interp]))

    scipy.misc.bytescale, scipy.misc.toimage, scipy.misc.fromimage = bytescale, toimage, fromimage
    scipy.misc.imresize, scipy.misc.imrotate = imresize, imrotate


def gen_pylib(R):
    from tests import inputs
    HumanPts, HumanAug, Evaluation, HumanAcc = R['HumanPts'], R['HumanAug'], R['Evaluation'], R['HumanAcc']
    out = {}
    # --- heat maps (a7), incl. hand-picked edge cases (Appendix A.1)
    special = np.array([[2.5, 2.5], [63.9, 63.9], [64.0, 64.0], [64.5, 10.0], [0.0, 0.0], [-1.0, 5.0],
                        [0.4, 0.4], [3.0, 61.2], [32.0, 32.0], [10.7, 20.2], [1e-3, 30.0], [63.0, 0.5],
                        [5.0, 64.0], [5.0, 64.01], [33.99, 2.01], [60.5, 60.5]])
    hm, valid = HumanPts.pts2heatmap(special.copy(), [64, 64], sigma=1)
    out['hm_special_pts'] = special
    out['hm_special'] = hm.astype(np.float32)
    out['hm_special_valid'] = valid
    pts = inputs.heat_pts(11, 4)
    hms = np.stack([HumanPts.pts2heatmap(pts[i].copy(), [64, 64], sigma=1)[0] for i in range(4)])
    out['hm_pts'] = pts
    out['hm_rand'] = hms[:2].astype(np.float32)          # first two samples stored in full
    out['hm_rand_digest'] = np.array([[h.sum(), (h * h).sum(), h.max()] for h in hms.reshape(-1, 64, 64)])
    # --- transforms (a8)
    c, s, r, gpts, norm = inputs.person_meta(12, 6)
    out['tf_c'], out['tf_s'], out['tf_r'], out['tf_pts'], out['tf_norm'] = c, s, r, gpts, norm
    out['tf_T256'] = np.stack([HumanAug.GetTransform(c[i], s[i], r[i], 256, 200) for i in range(6)])
    out['tf_T64'] = np.stack([HumanAug.GetTransform(c[i], s[i], r[i], 64, 200) for i in range(6)])
    out['tf_pts64'] = np.stack([HumanAug.TransformPts(gpts[i], c[i], s[i], r[i], 64, 200) for i in range(6)])
    out['tf_pts64_eval_inv'] = np.stack([
        Evaluation.TransformPts(out['tf_pts64'][i] + 1, c[i], s[i], r[i], 64, 200, invert=1) for i in range(6)])
    sh = np.stack([HumanAug.shufflelr(t(gpts[i].copy()), width=1280, dataset='mpii').numpy() for i in range(6)])
    out['tf_shufflelr'] = sh
    # --- argmax / PCKh (a11-a15) on heat maps built from the transformed joints
    n = 6
    tp = out['tf_pts64'].copy()
    tp[gpts[..., 0] <= 0] = 0
    tgt = np.stack([HumanPts.pts2heatmap(tp[i].copy(), [64, 64], sigma=1)[0] for i in range(n)]).astype(np.float32)
    pred = inputs.noisy_heatmaps(13, tgt, noise=0.2)
    pred[0, 3] = -1.0          # all-negative map -> zero prediction (Evaluation.py:21-22)
    # inputs are regenerated from seeds by the tests (tests/inputs.py); only a checksum is stored
    out['ev_pred_sum'] = np.array([pred.astype(np.float64).sum(), tgt.astype(np.float64).sum()])
    out['ev_get_preds'] = Evaluation.get_preds(t(pred)).numpy()
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]
    out['ev_accuracy'] = Evaluation.accuracy(t(pred), t(tgt), idx).numpy()
    cT, sT, rT = t(c).float(), t(s).float().view(n, 1), t(r).float().view(n, 1)
    out['ev_final_preds'] = Evaluation.final_preds(t(pred), cT, sT, [64, 64], rT).numpy()
    out['ev_acc_origin'] = Evaluation.accuracy_origin_res(t(pred), cT, sT, [64, 64], t(gpts).float(),
                                                           t(norm).float(), rT).numpy()
    out['ev_per_person'] = Evaluation.per_person_pckh(t(pred), t(tgt), cT, sT, [64, 64], t(gpts).float(),
                                                      t(norm).float(), rT).numpy()
    pp = out['ev_get_preds']
    gp = Evaluation.get_preds(t(tgt)).numpy()
    out['acc_approx_pckh'] = np.array(HumanAcc.approx_PCKh(t(pp), t(gp), idx, 64), dtype=np.float64)
    # the rest of the HumanAcc family (pylib/HumanAcc.py:46-308)
    import contextlib, io
    avg, per = HumanAcc.approx_PCKh_per(t(pp), t(gp), idx, 64)
    out['acc_per_avg'] = np.array(float(avg)); out['acc_per'] = per.numpy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        HumanAcc.PCKh(t(pp), t(gp), t(norm).float() / 20)
    out['acc_pckh_print'] = np.array([float(l.split(':')[1]) for l in buf.getvalue().strip().splitlines()])
    out['acc_samples'] = HumanAcc.approx_PCKh_samples(t(pp), t(gp), 64).numpy()
    out['acc_correct'] = HumanAcc.correct_predicted_joints(t(pp), t(gp), 64).numpy()
    out['acc_correct_orig'] = HumanAcc.correct_predicted_joints_original_resolution(t(pp), t(gp), 2.5).numpy()
    out['acc_dist_to_grnd'] = HumanAcc.predicted_joints_dist_to_grnd(t(pp), t(gp), 64).numpy()
    out['flip_maps'] = HumanAug.shuffle_channels_for_horizontal_flipping(HumanAug.flip_channels(t(pred[:1].copy()))).numpy()
    # --- loss (a6)
    w = inputs.rng(14).random(pred.shape, dtype=np.float32) + 0.5
    out['l2_weighted'] = np.array(float(R['Criterion'].weighted_L2(t(pred), t(tgt), t(w))))
    out['l2_unit'] = np.array(float(R['Criterion'].weighted_L2(t(pred), t(tgt), torch.ones(1))))
    # --- agent reward shaping (a18)
    g = inputs.rng(15)
    logits = g.normal(0, 1.5, (12, 7)).astype(np.float32)
    p = torch.softmax(t(logits), 1)
    p[3] = torch.tensor([0.9, 0.02, 0.02, 0.02, 0.02, 0.01, 0.01])      # forces the clamp branch
    p[4] = torch.tensor([0.0, 0.0, 0.5, 0.5, 0.0, 0.0, 0.0])
    ind = t(g.integers(0, 7, (12, 1)))
    ind[3, 0] = 0
    ind[4, 0] = 0
    pk_reg = t(g.random(12).astype(np.float32))
    pk_ag = t(g.random(12).astype(np.float32))
    pk_ag[5] = pk_reg[5]
    out['gg_p'] = p.numpy(); out['gg_idx'] = ind.numpy()
    out['gg_reg'] = pk_reg.numpy(); out['gg_agent'] = pk_ag.numpy()
    out['gg_out'] = R['util'].gen_groundtruth(p, ind, pk_reg, pk_ag).numpy()
    np.savez_compressed(os.path.join(OUT, 'pylib.npz'), **out)
    print('pylib.npz', len(out), 'arrays')


def digest(tensors):
    """per-tensor [sum, l2, first, last] -- compact fingerprint of a list of tensors."""
    return np.array([[float(x.sum()), float(x.norm()), float(x.flatten()[0]), float(x.flatten()[-1])]
                     for x in tensors], dtype=np.float64)


def gen_nets(R):
    from tests import inputs
    from oracle.model import deterministic_fill_
    from oracle import pylib as opl
    M = R['M']
    torch.set_num_threads(8)
    # --- one residual block, fwd + bwd (a1)
    blk = M._Residual(32, 32)
    deterministic_fill_(blk, seed=21)
    blk.train()
    x = t(inputs.rng(22).standard_normal((2, 32, 8, 8)).astype(np.float32)).requires_grad_(True)
    y = blk(x)
    gy = t(inputs.rng(23).standard_normal((2, 32, 8, 8)).astype(np.float32))
    y.backward(gy)
    np.savez_compressed(os.path.join(OUT, 'residual.npz'), y=y.detach().numpy(), dx=x.grad.numpy(),
                        grads=np.concatenate([p.grad.flatten().numpy() for p in blk.parameters()]),
                        running=np.concatenate([b.flatten().float().numpy() for b in blk.buffers()]))
    # --- small 1-stack net: every gradient and the post-RMSprop parameters (a2,a3,a6,a19)
    # hg_s1c128: a width the HIP engine runs (chan % 128 == 0): the engine is compared with THIS reference output directly
    for tag, (stacks, chan, seed) in {'hg_s1c8': (1, 8, 31), 'hg_s2c16': (2, 16, 32), 'hg_s1c128': (1, 128, 33)}.items():
        net = M.create_hg(num_stacks=stacks, num_modules=1, num_classes=16, chan=chan)
        deterministic_fill_(net, seed=seed)
        net.train()
        img = t(inputs.images(seed + 100, 2, 128))
        pts = inputs.heat_pts(seed + 200, 2, res=32)
        heat = t(inputs.heatmaps_from_pts(pts, res=32))
        assert np.array_equal(heat.numpy(), np.stack(
            [R['HumanPts'].pts2heatmap(pts[i].copy(), [32, 32])[0] for i in range(2)]).astype(np.float32))
        opt = torch.optim.RMSprop(net.parameters(), lr=2.5e-4, alpha=0.99, eps=1e-8, momentum=0, weight_decay=0)
        out = net(img)
        loss = 0
        for o in out:
            d = (o - heat) ** 2
            loss = loss + d.sum() / d.numel()
        opt.zero_grad()
        loss.backward()
        grads = [p.grad.clone() for p in net.parameters()]
        opt.step()
        acc = R['Evaluation'].accuracy(out[-1].detach(), heat, [0, 1, 2, 3, 4, 5, 10, 11, 14, 15])
        rec = dict(out=np.stack([o.detach().numpy() for o in out]), loss=np.array(float(loss)),
                   grad_digest=digest(grads), param_digest=digest([p.detach() for p in net.parameters()]),
                   buf_digest=digest([b.float() for b in net.buffers()]), acc=acc.numpy(),
                   nparams=np.array(sum(p.numel() for p in net.parameters())))
        if tag == 'hg_s1c8':
            rec['grads'] = np.concatenate([g.flatten().numpy() for g in grads])
        if tag == 'hg_s1c128':                   # the output layers' gradients in full (the engine is compared with them)
            names = [n for n, _ in net.named_parameters()]
            rec['head_names'] = np.array([n for n in names if n.startswith('out_conv.0.') or n.startswith('linear.0.1.')])
            rec['head_grads'] = np.concatenate([g.flatten().numpy() for n, g in zip(names, grads)
                                                if n.startswith('out_conv.0.') or n.startswith('linear.0.1.')])
        # eval-mode forward with the updated running stats
        net.eval()
        with torch.no_grad():
            rec['out_eval'] = np.stack([o.numpy() for o in net(img)])
        np.savez_compressed(os.path.join(OUT, tag + '.npz'), **rec)
        print(tag, rec['nparams'], 'params, loss', float(loss))
    # --- agent: half-hourglass logits, KL loss, agent gradients only (a4,a5,a17)
    net = M.create_hg(num_stacks=2, num_modules=1, num_classes=16, chan=16)
    asn = M.create_asn(chan_in=16, chan_out=16, scale_num=7, rotation_num=7, is_aug=True)
    deterministic_fill_(net, seed=41)
    deterministic_fill_(asn, seed=42)
    net.eval()
    asn.train()
    img = t(inputs.images(141, 2, 256))
    ls, lr = net(img, asn, is_half_hg=True, is_aug=True)
    ps, pr = torch.softmax(ls, 1), torch.softmax(lr, 1)
    g = inputs.rng(43)
    idx_s, idx_r = t(g.integers(0, 7, (2, 1))), t(g.integers(0, 7, (2, 1)))
    reg, ag = t(g.random(2).astype(np.float32)), t(g.random(2).astype(np.float32))
    gs = R['util'].gen_groundtruth(ps, idx_s, reg, ag)
    gr = R['util'].gen_groundtruth(pr, idx_r, ag, reg)
    import torch.nn.functional as F
    loss = F.kl_div(torch.log(ps + 1e-7), gs, reduction='mean') * 7 + \
        F.kl_div(torch.log(pr + 1e-7), gr, reduction='mean') * 7       # torch-0.3 default = element mean
    net.zero_grad(); asn.zero_grad()
    loss.backward()
    assert all(p.grad is None for p in net.parameters())               # features are detached (:161-162)
    np.savez_compressed(os.path.join(OUT, 'asn_c16.npz'), logits_s=ls.detach().numpy(), logits_r=lr.detach().numpy(),
                        idx_s=idx_s.numpy(), idx_r=idx_r.numpy(), reg=reg.numpy(), ag=ag.numpy(),
                        gs=gs.numpy(), gr=gr.numpy(), loss=np.array(float(loss)),
                        grad_digest=digest([p.grad for p in asn.parameters()]),
                        nparams=np.array(sum(p.numel() for p in asn.parameters())))
    print('asn', float(loss))
    # full-size parameter counts: 6 570 784 / 2 577 934
    n_hg = sum(p.numel() for p in M.create_hg(2, 1, 16, 256).parameters())
    n_asn = sum(p.numel() for p in M.create_asn(256, 256, 7, 7, is_aug=True).parameters())
    keys = list(M.create_hg(2, 1, 16, 256).state_dict().keys())
    np.savez_compressed(os.path.join(OUT, 'census.npz'), n_hg=np.array(n_hg), n_asn=np.array(n_asn),
                        hg_keys=np.array(keys))
    print('census', n_hg, n_asn, len(keys))


def gen_dropout(R):
    """Occlusion (dropout) agent branch, SURVEY.md section 8f rank 4 (models/asn_stacked_hg.py:79-136,172-190,308-321,437-439):
    mask logits of the half hourglass, the reference's own draw under np.random.seed, the masked two-stack forward,
    pose-net gradients of the MSE loss through the masks, agent gradients for a given upstream gradient of the logits."""
    import warnings
    from tests import inputs
    from oracle.model import deterministic_fill_
    M = R['M']
    warnings.simplefilter('ignore')
    net = M.create_hg(num_stacks=2, num_modules=1, num_classes=16, chan=16)
    asn = M.create_asn(chan_in=16, chan_out=16, is_dropout=True)
    deterministic_fill_(net, seed=51)
    deterministic_fill_(asn, seed=52)
    net.train(); asn.train()
    img = t(inputs.images(151, 2, 256))
    pts = inputs.heat_pts(152, 2, res=64)
    heat = t(inputs.heatmaps_from_pts(pts, res=64))
    import copy
    net_half = copy.deepcopy(net); asn_half = copy.deepcopy(asn)
    half = net_half(img, asn_half, is_half_hg=True, is_dropout=True)
    np.random.seed(153)
    out, pred_mask, indexes = net(img, asn, is_dropout=True)
    assert torch.allclose(half, pred_mask)
    loss = 0
    for o in out:
        d = (o - heat) ** 2
        loss = loss + d.sum() / d.numel()
    net.zero_grad(); asn.zero_grad()
    loss.backward(retain_graph=True)
    assert all(p.grad is None for p in asn.parameters())            # the agent sees detached features, the masks are constants
    pose_grads = [p.grad.clone() for p in net.parameters()]
    gy = t(inputs.rng(154).standard_normal((2, 1, 4, 4)).astype(np.float32))
    pred_mask.backward(gy)
    # _dropout alone on a random tensor (:79-100)
    x = t(inputs.rng(155).standard_normal((2, 3, 16, 16)).astype(np.float32))
    masks = torch.ones(2, 1, 4, 4); masks[0, 0, 1, 2] = 0; masks[1, 0, 3, 0] = 0; masks[1, 0, 0, 0] = 0
    dx = net.hg[0]._dropout(x, masks)
    # the sampler under a fixed numpy seed, on sharper logits
    np.random.seed(156)
    lg = t(inputs.rng(157).normal(0, 2.0, (6, 1, 4, 4)).astype(np.float32))
    smask, sidx = net.hg[0]._sample_mask(lg)
    np.savez_compressed(os.path.join(OUT, 'dropout_c16.npz'), pred_mask=pred_mask.detach().numpy(), indexes=indexes.numpy(),
                        out=np.stack([o.detach().numpy() for o in out]), loss=np.array(float(loss)),
                        pose_grad_digest=digest(pose_grads), asn_grad_digest=digest([p.grad for p in asn.parameters()]),
                        asn_keys=np.array(list(asn.state_dict().keys())),
                        nparams=np.array(sum(p.numel() for p in asn.parameters())),
                        drop_x_out=dx.numpy(), drop_masks=masks.numpy(),
                        sample_masks=smask.numpy(), sample_idx=sidx.numpy())
    print('dropout', float(loss), indexes.tolist())


def gen_crop(R):
    """Row a9: the reference's own crop (pylib/HumanAug.py:117-176) behind the reference's own pre-processing
    (data/mpii_for_mpii.py:114-135: load as fp32 / 255, mirror, colour gain + clamp) on 720x1280 frames at res 256, over
    the scipy.misc shim above and the real Pillow.  Stored per case: every 4th pixel of the crop (rows 1::4, columns 2::4), per-channel byte sums of the
    full crop, and the full crop for three of the cases."""
    from tests import inputs
    sys.path.insert(0, TMP)
    from utils import imutils
    HumanAug = R['HumanAug']
    out = {}
    frames = {}
    for i, (kind, c0, s0, r0, flip, gain, neutral) in enumerate(inputs.WARP_CASES):
        if kind not in frames:
            frames[kind] = inputs.warp_frame(kind)
        img = imutils.im_to_torch(frames[kind].copy())                    # load_image: uint8 HWC -> fp32 CHW / 255
        c = torch.Tensor(list(c0))
        s = torch.Tensor([s0])
        if flip:
            img = torch.from_numpy(HumanAug.fliplr(img.numpy())).float()
            c[0] = img.size(2) - c[0]
        for k in range(3):
            img[k, :, :].mul_(gain[k]).clamp_(0, 1)
        inp = HumanAug.crop(imutils.im_to_numpy(img), c.numpy(), s.numpy(), r0, 256, 200)
        assert inp.dtype == np.uint8 and inp.shape == (256, 256, 3)
        out['crop%02d_sub' % i] = inp[1::4, 2::4].copy()
        out['crop%02d_sums' % i] = np.array([inp[..., k].astype(np.int64).sum() for k in range(3)]
                                            + [(inp[..., k].astype(np.int64) ** 2).sum() for k in range(3)])
        if i in (0, 5, 13):
            out['crop%02d_full' % i] = inp
    out['frame_sums'] = np.array([[int(frames[k].astype(np.int64).sum())] for k in sorted(frames)])
    np.savez_compressed(os.path.join(OUT, 'crop.npz'), **out)
    print('crop.npz', len(out), 'arrays', os.path.getsize(os.path.join(OUT, 'crop.npz')) // 1024, 'KB')


def gen_dataset(R):
    """Missing piece of round 2: ONE WHOLE dataset sample, JSON -> (inp, heatmap, c, s, r, pts, normalizer), through the
    reference's own MPII.__getitem__ (data/mpii_for_mpii.py:83-163, train and val) and AGENT.__getitem__
    (data/joint_train_s_r_agent.py:98-177, bins given / separate_s_r) with np.random seeded, on the 3-person set of
    tests/inputs.py written to /tmp as JSON + PNG.  Pins the COMPOSITION: the call order of the draws, `c.x = W - c.x`
    before the crop, shufflelr before TransformPts, the `pts <= 0` zeroing, heat maps from the transformed joints.
    torchvision / matplotlib are imported (never used) by the reference's data / utils modules and are absent here: empty
    stand-in modules satisfy the import; scipy.misc.imread = PIL open + convert('RGB') (scipy 0.19 pilutil.imread)."""
    import types
    import scipy.misc
    from PIL import Image
    from tests import inputs
    for name in ('torchvision', 'torchvision.transforms', 'matplotlib', 'matplotlib.pyplot'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']
    scipy.misc.imread = lambda path, mode='RGB': np.array(Image.open(path).convert(mode))
    sys.path.insert(0, TMP)
    from data import mpii_for_mpii as D
    from data import joint_train_s_r_agent as A
    folder = '/tmp/ref3_goldens_dataset'
    if os.path.isdir(folder):
        shutil.rmtree(folder)
    jpath, frames, anno = inputs.write_dataset(folder)
    out = {'frame_sums': np.array([int(f.astype(np.int64).sum()) for f in frames])}

    def pack(tag, inp, heat, c, s, r, pts, normalizer):
        b = np.rint(inp.numpy() * 255.0).astype(np.uint8)                    # the crop's bytes (im_to_torch divided by 255)
        assert np.array_equal((b.astype(np.float32) / 255).astype(np.float32), inp.numpy()), tag
        out[tag + '_inp_sub'] = b[:, 1::4, 2::4].copy()
        out[tag + '_inp_sums'] = np.array([b[k].astype(np.int64).sum() for k in range(3)] + [(b[k].astype(np.int64) ** 2).sum() for k in range(3)])
        out[tag + '_heat'] = heat.numpy()
        out[tag + '_c'] = np.asarray(c.numpy(), dtype=np.float32)
        out[tag + '_s'] = np.asarray(s.numpy(), dtype=np.float32)
        out[tag + '_r'] = np.asarray(r.numpy(), dtype=np.float32)
        out[tag + '_pts'] = np.asarray(pts.numpy(), dtype=np.float32)
        out[tag + '_normalizer'] = np.float64(normalizer)
        return b

    train = D.MPII(jpath, folder, is_train=True)
    val = D.MPII(jpath, folder, is_train=False)
    assert len(train) == 2 and len(val) == 1
    seeds = [11, 12, 13, 14, 15, 16]
    flips, rots = [], []
    for index in (0, 1):
        for seed in seeds:
            np.random.seed(seed)
            inp, heat, c, s, r, pts, normalizer = train[index]
            d = inputs.legacy_draws(seed)
            flips.append(d[3] <= 0.5); rots.append(float(r[0]) != 0)
            b = pack('train%d_seed%d' % (index, seed), inp, heat, c, s, r, pts, normalizer)
            if index == 0 and seed == seeds[0]:
                out['train0_seed%d_inp_full' % seed] = b
    assert any(flips) and not all(flips) and any(rots) and not all(rots), (flips, rots)
    inp, heat, c, s, r, pts, normalizer, index = val[0]
    assert index == 0
    pack('val0', inp, heat, c, s, r, pts, normalizer)
    # the agent's dataset: bins given (separate_s_r False: flip + colour as in the regular law) ...
    ag = A.AGENT(jpath, folder, separate_s_r=False)
    ag.img_index_list, ag.scale_index_list, ag.rotation_index_list = [1, 0], [5, 1], [0, 6]
    for k in (0, 1):
        np.random.seed(21 + k)
        inp, heat, c, s, r, pts, normalizer, idx = ag[k]
        assert idx == ag.img_index_list[k]
        pack('agent%d' % k, inp, heat, c, s, r, pts, normalizer)
    out['agent_img_index'] = np.array(ag.img_index_list); out['agent_scale_index'] = np.array(ag.scale_index_list)
    out['agent_rot_index'] = np.array(ag.rotation_index_list)
    # ... and separate_s_r (scale-only crop, rotation-only crop; no flip, no colour), bins given
    ag2 = A.AGENT(jpath, folder, separate_s_r=True)
    ag2.img_index_list, ag2.scale_index_list, ag2.rotation_index_list = [0], [6], [2]
    np.random.seed(31)
    img_list, heat_list, c_list, s_list, r_list, pts_list, norm_list, idx = ag2[0]
    for k, name in enumerate(('sep_scale', 'sep_rot')):
        pack(name, img_list[k], heat_list[k], c_list[k], s_list[k], r_list[k], pts_list[k], norm_list[k])
    np.savez_compressed(os.path.join(OUT, 'dataset.npz'), **out)
    print('dataset.npz', len(out), 'arrays', os.path.getsize(os.path.join(OUT, 'dataset.npz')) // 1024, 'KB; flips', flips, 'rotated', rots)


if __name__ == '__main__':
    R = load_ref()
    if len(sys.argv) > 1 and sys.argv[1] == 'crop':
        gen_crop(R)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'dataset':
        gen_dataset(R)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'nets':
        gen_nets(R)
        sys.exit(0)
    gen_pylib(R)
    gen_nets(R)
    gen_dropout(R)
    gen_crop(R)
    gen_dataset(R)
