"""GPU parity of the network kernels (bf16 storage, fp32 accumulation) against the CPU oracle
(fp32 PyTorch restatement pinned to the reference by tests/test_oracle_golden.py).

Tolerances (stated per check): activations are stored in bf16 (8 mantissa bits, rel. step 2^-8 = 3.9e-3)
between every convolution, so after ~100 layers outputs agree to a few 1e-2 of their RMS; gradients
likewise.  rel_rms(a, b) = rms(a - b) / rms(b)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import pylib as opl
from oracle import step as ostep
from tests import inputs

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def rel_rms(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def cosine(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def test_layout_roundtrip():
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    x = t(inputs.rng(1).standard_normal((2, 64, 5, 7)).astype(np.float32)).cuda()
    nhwc = torch.empty((2, 5, 7, 64), dtype=torch.bfloat16, device='cuda')
    back = torch.empty_like(x)
    check(lib().pa_nchw_to_nhwc(ptr(x), ptr(nhwc), 2, 64, 5, 7, stream()))
    check(lib().pa_nhwc_to_nchw(ptr(nhwc), ptr(back), 2, 64, 5, 7, stream()))
    assert torch.equal(nhwc.float().permute(0, 3, 1, 2), x.bfloat16().float())
    assert torch.equal(back, x.bfloat16().float())


@pytest.mark.parametrize('C_,H,B', [(128, 16, 2), (256, 8, 3), (128, 4, 1), (256, 64, 2), (256, 64, 6), (128, 64, 8), (256, 32, 8)])
def test_residual_block_fwd_bwd(C_, H, B):
    """_Residual(C, C): 1x1 -> 3x3 -> 1x1 (+identity) with train-mode BatchNorm; forward, input gradient,
    every parameter gradient and the running statistics.  Ragged M (B*H*H not a multiple of the tile)
    is covered by (128, 4, 1) -> M = 16 and (256, 8, 3) -> M = 192."""
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    blk = om.Residual(C_, C_)
    om.deterministic_fill_(blk, seed=3)
    blk.train()
    x = t(inputs.rng(4).standard_normal((B, C_, H, H)).astype(np.float32))
    x = torch.relu(x)                                    # block inputs are post-ReLU activations in the net
    dy = t(inputs.rng(5).standard_normal((B, C_, H, H)).astype(np.float32))
    xr = x.bfloat16().float().requires_grad_(True)
    y = blk(xr)
    y.backward(dy.bfloat16().float())
    sd = blk.state_dict()
    params = torch.cat([p.detach().flatten() for p in blk.parameters()]).cuda()
    bufs = torch.cat([b.flatten().float() for n, b in blk.named_buffers() if 'num_batches' not in n])
    # buffers before the step (the oracle already updated its own): re-create the initial ones
    blk0 = om.Residual(C_, C_); om.deterministic_fill_(blk0, seed=3)
    bufs0 = torch.cat([b.flatten().float() for n, b in blk0.named_buffers() if 'num_batches' not in n]).cuda()
    nws = lib().pa_residual_workspace_bytes(B, H, H, C_)
    ws = torch.zeros(nws, dtype=torch.uint8, device='cuda')
    yd = torch.empty_like(x).cuda(); dxd = torch.empty_like(x).cuda(); gd = torch.zeros_like(params)
    xd, dyd = x.cuda(), dy.cuda()          # keep the device copies alive across the call
    check(lib().pa_residual_fwd_bwd(ptr(xd), ptr(dyd), ptr(params), ptr(yd), ptr(dxd), ptr(gd), ptr(bufs0),
                                    B, C_, H, H, ptr(ws), stream()), 'pa_residual_fwd_bwd')
    # (1) against the fp32 oracle: forward to bf16 storage precision; gradients of this test are
    #     random-walk sums (dy is white noise), so the few ReLU masks that flip under bf16 rounding of
    #     the pre-activations show up as a few % -- bounded here, explained by (2)
    assert rel_rms(yd.cpu(), y.detach()) < 1e-2, 'forward'
    assert rel_rms(dxd.cpu(), xr.grad) < 0.1 and cosine(dxd.cpu(), xr.grad) > 0.995, 'input gradient'
    # (2) against the bf16-storage emulation of the same block (tests/bf16_emul.py): same rounding
    #     points, so only accumulation order differs -> tight
    from tests import bf16_emul
    ye, dxe, ge = bf16_emul.residual_fwd_bwd(blk0, x, dy)
    assert rel_rms(yd.cpu(), ye) < 3e-3, 'forward vs bf16 emulation'
    assert rel_rms(dxd.cpu(), dxe) < 1.5e-2 and cosine(dxd.cpu(), dxe) > 0.9998, 'input gradient vs bf16 emulation'
    assert rel_rms(dxe, xr.grad) < 0.1, 'emulation vs fp32 oracle (documents the bf16 storage error)'
    off = 0
    for name, p in blk.named_parameters():
        n = p.numel()
        got = gd[off:off + n].cpu().view_as(p)
        off += n
        if name.endswith('bias') and 'bn' not in name:
            # bias in front of a BatchNorm: the true gradient is 0 (the reference's value is rounding noise)
            assert float(got.abs().max()) == 0.0 and float(p.grad.abs().max()) < 2e-2 * float(blk.bn3.bias.grad.abs().max()), name
            continue
        assert rel_rms(got, p.grad) < 0.15 and cosine(got, p.grad) > 0.99, name
        assert rel_rms(got, ge[name]) < 2e-2 and cosine(got, ge[name]) > 0.9998, name + ' vs bf16 emulation'
    assert rel_rms(bufs0.cpu(), bufs) < 1e-2, 'running statistics'


def _hg_pair(stacks, chan, B, res, seed):
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    ref = om.create_hg(stacks, 1, 16, chan)
    om.deterministic_fill_(ref, seed=seed)
    net = create_hg(stacks, 1, 16, chan, res=res, default_batch=B)
    keys = list(net.state_dict().keys())
    assert keys == list(ref.state_dict().keys()), 'state_dict names/order must match the reference module tree'
    for (k, a), (k2, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert tuple(a.shape) == tuple(b.shape), k
    net.load_state_dict(ref.state_dict())
    return ref, net


def test_hourglass_forward_backward_vs_oracle():
    """2-stack hourglass, chan 128, B=2, 128x128 input: outputs (train-mode BN), loss, every gradient."""
    torch.set_num_threads(8)
    B, res, chan = 2, 128, 128
    ref, net = _hg_pair(2, chan, B, res, seed=7)
    img = t(inputs.images(8, B, res))
    pts = inputs.heat_pts(9, B, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    ref.train(); net.train()
    out_ref, loss_ref = ostep.pose_loss_and_grads(ref, img, heat)
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    # End-to-end the untrained net is chaotic (tests/test_gpu_local.py explains and checks every node
    # tightly in place): the bf16-storage emulation of the ORACLE itself differs from the fp32 oracle by
    # the same amount as the engine does.  So: loss to 1 %, heat maps to 15 % of their RMS, and the
    # engine must be as close to the fp32 oracle as the bf16 emulation of the oracle is (factor 1.5).
    from tests import bf16_emul
    import copy
    outs_e = bf16_emul.emul_hourglass_net(copy.deepcopy(ref), img)     # a copy: keep ref's running stats untouched
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-2
    for o, r, e in zip(outs, out_ref, outs_e):
        err_hip, err_emul = rel_rms(o.cpu(), r.detach()), rel_rms(e.detach(), r.detach())
        assert err_hip < 0.15 and err_hip < 1.5 * err_emul + 1e-2, (err_hip, err_emul)
    # the output layers' gradients are still well conditioned
    gref = dict(ref.named_parameters())
    for name, g in net.named_grads():
        if name.startswith('out_conv.%d.' % 1) or name.startswith('linear.1.1.'):
            assert rel_rms(g.cpu(), gref[name].grad) < 5e-2 and cosine(g.cpu(), gref[name].grad) > 0.998, name
    # running statistics after one train-mode forward
    for (k, a), (k2, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert rel_rms(a.cpu(), b) < 6e-2, k       # low-resolution levels: few samples, chaotic (see above)
    # eval-mode forward with those running statistics + device PCK == oracle PCK on the device's own maps
    ref.eval(); net.eval()
    with torch.no_grad():
        oe_ref = ref(img)
    oe = net(img.cuda(), pts=t(pts).cuda())
    assert rel_rms(oe[-1].cpu(), oe_ref[-1]) < 6e-2
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]
    acc_dev = net.accuracy(idx).cpu().numpy()
    acc_ref = opl.accuracy(oe[-1].cpu(), heat, idx).numpy()
    assert np.allclose(acc_dev, acc_ref, atol=1e-4)


def test_training_steps_track_the_oracle():
    """5 RMSprop steps on a fixed batch: the HIP engine's loss curve follows the fp32 oracle's."""
    from pose_adv_aug_amd.utils.optim import RMSprop
    torch.set_num_threads(8)
    B, res, chan = 2, 128, 128
    ref, net = _hg_pair(1, chan, B, res, seed=17)
    img = t(inputs.images(18, B, res))
    pts = inputs.heat_pts(19, B, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    opt_ref = ostep.make_optimizer(ref)
    opt = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8)
    ref.train(); net.train()
    lr_, ld_ = [], []
    for _ in range(5):
        _, l = ostep.pose_loss_and_grads(ref, img, heat)
        opt_ref.step()
        lr_.append(float(l))
        l2, _ = net.loss_and_backward(img.cuda(), t(pts).cuda())
        opt.step()
        ld_.append(float(l2))
    # same first loss (to bf16), both decrease, and the trajectories stay together (RMSprop's first steps
    # are ~ +-10 lr * sign(g) on every weight, so the untrained net's chaos shows up quickly: 25 %)
    assert abs(ld_[0] - lr_[0]) / lr_[0] < 1e-2, (ld_, lr_)
    assert ld_[-1] < ld_[0] and lr_[-1] < lr_[0], (ld_, lr_)
    for a, b in zip(ld_, lr_):
        assert abs(a - b) / b < 0.25, (ld_, lr_)


def test_full_size_config_runs_and_is_finite():
    """BASELINE config 2 shape: 2-stack, chan 256, B=24, 256x256 -- one train step, finite loss/grads,
    loss equals the loss recomputed from the returned heat maps (size-independent consistency)."""
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd import pylib
    net = create_hg(2, 1, 16, 256)
    net.reset_parameters(seed=0)
    img = t(inputs.images(28, 24, 256)).cuda()
    pts = t(inputs.heat_pts(29, 24)).cuda()
    net.train()
    loss, outs = net.loss_and_backward(img, pts, want_outputs=True)
    tgt = pylib.HumanPts.pts2heatmap_batch(pts, 64, 64)
    recomputed = sum(float(pylib.Criterion.weighted_L2(o, tgt, 1)) for o in outs)
    assert np.isfinite(float(loss)) and abs(float(loss) - recomputed) / recomputed < 1e-4
    assert bool(torch.isfinite(net.flat_grads).all())
    assert net.num_params() == 6570784


def test_c2_size_step_matches_oracle():
    """BASELINE configs[1] at FULL size (2-stack, chan 256, B=24, 256x256), one training step against the fp32 oracle on
    the same weights and inputs: loss (1 %), the last stack's output-layer gradients (out_conv.1, linear.1.1: rel-rms 5 %,
    cosine .998 -- deeper gradients of an untrained net are chaotic under bf16 storage, tests/test_gpu_local.py pins them
    node by node), running statistics of the stem, and the IN-ENGINE metrics of the timed region -- pa_hg_accuracy and
    pa_hg_pckh (accuracy_origin_res AND per_person_pckh, the joint loop's reward) -- against the oracle's
    pylib/Evaluation.py restatement on the engine's own heat maps to 1e-4."""
    torch.set_num_threads(max(8, torch.get_num_threads()))
    B, res, chan = 24, 256, 256
    ref, net = _hg_pair(2, chan, B, res, seed=11)
    img = t(inputs.images(41, B, res))
    c, s, r, gpts, norm = inputs.person_meta(42, B)
    pts = np.stack([opl.transform_pts(gpts[i], c[i], s[i], r[i], 64) for i in range(B)])
    pts[gpts[..., 0] <= 0] = 0
    heat = t(inputs.heatmaps_from_pts(pts, res=64))
    ref.train(); net.train()
    out_ref, loss_ref = ostep.pose_loss_and_grads(ref, img, heat)
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-2, (float(loss), float(loss_ref))
    gref = dict(ref.named_parameters())
    checked = 0
    for name, g in net.named_grads():
        if name.startswith('out_conv.1.') or name.startswith('linear.1.1.'):
            assert rel_rms(g.cpu(), gref[name].grad) < 5e-2 and cosine(g.cpu(), gref[name].grad) > 0.998, name
            checked += 1
    assert checked == 4
    sd, sr = net.state_dict(), ref.state_dict()
    for k in ('bn1.running_mean', 'bn1.running_var', 'residual1.bn1.running_mean'):
        assert rel_rms(sd[k].cpu(), sr[k]) < 2e-2, k
    # metrics of the timed region, computed by the engine on its own heat maps vs the oracle on the same maps
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]
    om_ = outs[-1].cpu()
    assert np.allclose(net.accuracy(idx).cpu().numpy(), opl.accuracy(om_, heat, idx).numpy(), atol=1e-4)
    cT, sT, rT = t(c).float(), t(s).float().view(B, 1), t(r).float().view(B, 1)
    acc, person = net.pckh_origin_res(cT.cuda(), t(s).float().cuda(), t(r).float().cuda(), t(gpts).float().cuda(), t(norm).float().cuda(),
                                      per_person=True)
    acc_ref = opl.accuracy_origin_res(om_, cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)
    person_ref = opl.per_person_pckh(om_, heat, cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)
    assert np.allclose(acc.cpu().numpy(), acc_ref.numpy(), atol=1e-4), (acc.cpu().numpy(), acc_ref.numpy())
    assert np.allclose(person.cpu().numpy(), person_ref.numpy(), atol=1e-4)
    # ... and on trained-looking maps (an untrained net scores ~0 everywhere): the oracle's target + noise fed through the same entry
    from pose_adv_aug_amd import pylib
    noisy = t(inputs.noisy_heatmaps(43, heat.numpy(), noise=0.2))
    a2 = pylib.Evaluation.accuracy_origin_res(noisy, cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).cpu().numpy()
    p2 = pylib.Evaluation.per_person_pckh(noisy, heat, cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).cpu().numpy()
    assert np.allclose(a2, opl.accuracy_origin_res(noisy, cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).numpy(), atol=1e-4)
    assert np.allclose(p2, opl.per_person_pckh(noisy, heat, cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT).numpy(), atol=1e-4)
    assert a2[0] > 0.3


def test_c1_plumbing_config_step_matches_oracle():
    """BASELINE configs[0] on the HIP path: stack-hg.py's 1-stack hourglass, chan 256, bs = 2, 256x256 (stack-hg.py:40-41 with
    num_stacks = 1) -- the reference's own CPU-runnable plumbing case, whose shapes no other GPU test visits (B = 2 makes the 4x4
    level a 32-pixel map at mid-width 128: generic kernels, one or two statistics rows).  One training step against the fp32 oracle
    on the same weights and inputs: loss 1 %, the head's gradients (out_conv.0, linear.0.1) 5 % / cosine .998, PCK of the engine
    equal to the oracle's on the engine's maps (1e-4)."""
    torch.set_num_threads(max(8, torch.get_num_threads()))
    B, res, chan = 2, 256, 256
    ref, net = _hg_pair(1, chan, B, res, seed=23)
    img = t(inputs.images(51, B, res))
    c, s, r, gpts, norm = inputs.person_meta(52, B)
    pts = np.stack([opl.transform_pts(gpts[i], c[i], s[i], r[i], 64) for i in range(B)])
    pts[gpts[..., 0] <= 0] = 0
    heat = t(inputs.heatmaps_from_pts(pts, res=64))
    ref.train(); net.train()
    out_ref, loss_ref = ostep.pose_loss_and_grads(ref, img, heat)
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-2, (float(loss), float(loss_ref))
    gref = dict(ref.named_parameters())
    checked = 0
    for name, g in net.named_grads():
        if name.startswith('out_conv.0.') or name.startswith('linear.0.1.'):
            assert rel_rms(g.cpu(), gref[name].grad) < 5e-2 and cosine(g.cpu(), gref[name].grad) > 0.998, name
            checked += 1
    assert checked == 4
    assert bool(torch.isfinite(net.flat_grads).all())
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]
    om_ = outs[-1].cpu()
    assert np.allclose(net.accuracy(idx).cpu().numpy(), opl.accuracy(om_, heat, idx).numpy(), atol=1e-4)
    cT, sT, rT = t(c).float(), t(s).float().view(B, 1), t(r).float().view(B, 1)
    acc = net.pckh_origin_res(cT.cuda(), t(s).float().cuda(), t(r).float().cuda(), t(gpts).float().cuda(), t(norm).float().cuda())
    acc = acc[0] if isinstance(acc, (tuple, list)) else acc
    acc_ref = opl.accuracy_origin_res(om_, cT, sT, [64, 64], t(gpts).float(), t(norm).float(), rT)
    assert np.allclose(acc.cpu().numpy(), acc_ref.numpy(), atol=1e-4), (acc.cpu().numpy(), acc_ref.numpy())


def test_eight_stack_384_config_matches_oracle_loss():
    """SURVEY.md config C5 shape (8-stack, 384x384 -> 96x96 maps; here B=2): map sizes 96, 48, 24, 12, 6 exercise the
    halo-tile kernels (96, 48) AND the generic ones (24, 12, 6: not multiples of 8x16).  Loss vs the fp32 oracle,
    loss consistency with the returned heat maps, finite gradients, parameter count of the reference module tree."""
    torch.set_num_threads(8)
    B, res, chan = 2, 384, 256
    ref, net = _hg_pair(8, chan, B, res, seed=5)
    img = t(inputs.images(31, B, res))
    pts = inputs.heat_pts(32, B, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    ref.train(); net.train()
    with torch.no_grad():
        out_ref = ref(img)
    loss_ref = float(opl.stack_mse(out_ref, heat))
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    assert len(outs) == 8 and tuple(outs[-1].shape) == (B, 16, 96, 96)
    assert abs(float(loss) - loss_ref) / loss_ref < 2e-2, (float(loss), loss_ref)
    recomputed = sum(float(((o.cpu() - heat) ** 2).mean()) for o in outs)
    assert abs(float(loss) - recomputed) / recomputed < 1e-4
    assert bool(torch.isfinite(net.flat_grads).all()) and float(net.flat_grads.abs().max()) > 0
    assert net.num_params() == sum(p.numel() for p in ref.parameters())
    # first stack's heat map: 3 blocks + one hourglass deep, still well conditioned
    assert rel_rms(outs[0].cpu(), out_ref[0]) < 0.15


def test_training_is_bitwise_reproducible_across_runs_and_stream_modes():
    """Race screen for the whole step.  Every reduction of the engine is order-deterministic (per-workgroup partial
    rows, slab reduction, shuffle trees -- no float atomics on anything that feeds the gradients), so N training steps
    from the same state must give BITWISE identical parameters and running statistics from run to run, and with the
    engine's side streams (hourglass skip branches, weight-gradient stream) on or off.  A missing event / barrier shows
    up here as a mismatch."""
    from pose_adv_aug_amd import _lib
    from pose_adv_aug_amd.stack_hg import train_step
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    B, steps = 24, 4

    def run(multi):
        net = create_hg(2, 1, 16, 256, res=256, default_batch=B); net.reset_parameters(seed=0)
        opt = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8); aug = Augmenter(seed=1)
        batches = [DeviceBatch.synthetic(B, seed=k) for k in range(2)]
        net.train()
        _lib.check(_lib.lib().pa_net_set_multi_stream(net._net(B), 1 if multi else 0))
        for i in range(steps):
            train_step(net, opt, aug, batches[i % 2])
        torch.cuda.synchronize()
        return net.flat_params.clone(), net.flat_buffers.clone()
    a, b, c = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), 'run-to-run mismatch (race?)'
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]), 'side streams change the result (missing dependency?)'
    assert bool(torch.isfinite(a[0]).all())


@pytest.mark.parametrize('C_,H,B', [(128, 16, 2), (128, 64, 8), (256, 32, 8), (256, 8, 3)])
def test_residual_op_stays_inside_its_workspace(C_, H, B):
    """pa_residual_workspace_bytes must cover everything pa_residual_fwd_bwd lays out (the split counts of the weight
    gradients depend on the map size: an under-estimate once let the op write 3 MB past its workspace).  Guard regions of
    64 MB on both sides of the workspace stay untouched."""
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    L = lib()
    nws = L.pa_residual_workspace_bytes(B, H, H, C_)
    SL = 64 << 20
    big = torch.zeros(nws + 2 * SL, dtype=torch.uint8, device='cuda')
    big[:SL] = 0x5A; big[SL + nws:] = 0x5A
    blk = om.Residual(C_, C_); om.deterministic_fill_(blk, seed=3)
    params = torch.cat([p.detach().flatten() for p in blk.parameters()]).cuda()
    bufs = torch.cat([b.flatten().float() for n, b in blk.named_buffers() if 'num_batches' not in n]).cuda()
    x = torch.rand(B, C_, H, H, device='cuda'); dy = torch.randn(B, C_, H, H, device='cuda')
    yd = torch.empty_like(x); dxd = torch.empty_like(x); gd = torch.zeros_like(params)
    check(L.pa_residual_fwd_bwd(ptr(x), ptr(dy), ptr(params), ptr(yd), ptr(dxd), ptr(gd), ptr(bufs), B, C_, H, H,
                                C.c_void_p(big.data_ptr() + SL), stream()), 'pa_residual_fwd_bwd')
    torch.cuda.synchronize()
    assert bool((big[:SL] == 0x5A).all()) and bool((big[SL + nws:] == 0x5A).all())
    assert bool(torch.isfinite(dxd).all()) and bool(torch.isfinite(gd).all())


def test_network_step_stays_inside_its_workspace():
    """pa_net_workspace_bytes vs what pa_hg_forward / pa_hg_backward and the agent touch: 64 MB guard regions around the
    workspaces of a 2-stack pose net (chan 128, 256x256, B = 2) and of its agent stay untouched."""
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    L = lib()
    B, res, chan = 2, 256, 128
    SL = 64 << 20

    def bound(h):
        nws = L.pa_net_workspace_bytes(h)
        big = torch.zeros(nws + 2 * SL, dtype=torch.uint8, device='cuda')
        big[:SL] = 0x5A; big[SL + nws:] = 0x5A
        P = torch.rand(L.pa_net_param_floats(h), device='cuda') * 0.1
        G = torch.zeros_like(P)
        Bf = torch.ones(max(1, L.pa_net_buffer_floats(h)), device='cuda')
        check(L.pa_net_bind(h, ptr(P), ptr(G), ptr(Bf), C.c_void_p(big.data_ptr() + SL), stream()), 'pa_net_bind')
        return big, nws, (P, G, Bf)

    hp = L.pa_hg_create(2, 16, chan, B, res)
    ha = L.pa_asn_create(chan, 7, 7, B, res)
    bp, np_, keep_p = bound(hp)
    ba, na, keep_a = bound(ha)
    img = torch.rand(B, 3, res, res, device='cuda')
    pts = (torch.rand(B, 16, 2, device='cuda', dtype=torch.float64) * 60 + 2)
    loss = torch.empty(2, device='cuda')
    check(L.pa_hg_forward(hp, ptr(img), None, ptr(pts), 1, ptr(loss)), 'pa_hg_forward')
    check(L.pa_hg_backward(hp), 'pa_hg_backward')
    check(L.pa_hg_forward_half(hp, ptr(img), None, 1), 'pa_hg_forward_half')
    ls = torch.empty(B, 7, device='cuda'); lr = torch.empty(B, 7, device='cuda')
    check(L.pa_asn_forward(ha, hp, 1, ptr(ls), ptr(lr)), 'pa_asn_forward')
    t7 = torch.full((B, 7), 1.0 / 7, device='cuda')
    al = torch.zeros(1, device='cuda')
    check(L.pa_asn_backward(ha, hp, ptr(t7), ptr(t7), ptr(al)), 'pa_asn_backward')
    torch.cuda.synchronize()
    for big, nws in ((bp, np_), (ba, na)):
        assert bool((big[:SL] == 0x5A).all()) and bool((big[SL + nws:] == 0x5A).all())
    assert bool(torch.isfinite(loss).all()) and bool(torch.isfinite(keep_p[1]).all()) and bool(torch.isfinite(keep_a[1]).all())
    L.pa_net_destroy(hp); L.pa_net_destroy(ha)


def test_augmentation_one_step_ahead_changes_nothing():
    """stack_hg.train / bench.py enqueue the NEXT batch's augmentation on a side stream one step ahead (data.AugmentAhead):
    same draws, same crops, same parameters after 4 steps -- bitwise -- as augmenting in front of every step."""
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import AugmentAhead, Augmenter, DeviceBatch
    from pose_adv_aug_amd.stack_hg import train_step
    batches = [DeviceBatch.synthetic(4, seed=k) for k in range(2)]
    finals = []
    for mode in ('inline', 'ahead'):
        net = create_hg(1, 1, 16, 128, default_batch=4); net.reset_parameters(seed=3); net.train()
        opt = RMSprop(net, lr=2.5e-4)
        aug = Augmenter(seed=9)
        ahead = AugmentAhead(aug)
        if mode == 'ahead':
            ahead.start(batches[0])
        for i in range(4):
            data = None
            if mode == 'ahead':
                data = ahead.take()
                ahead.start(batches[(i + 1) % 2] if i < 3 else None)
            loss, _, _ = train_step(net, opt, aug, batches[i % 2], data=data)
        torch.cuda.synchronize()
        finals.append((net.flat_params.clone(), net.flat_buffers.clone(), float(loss)))
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])
    assert abs(finals[0][2] - finals[1][2]) < 1e-6 * abs(finals[0][2])        # (the reported loss is summed with one float atomic per workgroup)


@pytest.mark.parametrize('stacks,chan,B', [(1, 128, 4), (2, 256, 24)])
def test_hip_graph_replay_equals_eager_enqueue(stacks, chan, B):
    """pa_hg_train_step(use_graph=1): forward + backward captured once into a HIP graph (side / weight-gradient streams as
    graph branches) and replayed -- bitwise the parameters, running statistics and gradients of the eager launch sequence,
    over 4 steps on changing inputs (the capture happens at step 0, steps 1-3 are replays)."""
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd.stack_hg import train_step
    batches = [DeviceBatch.synthetic(B, seed=40 + k) for k in range(2)]
    finals = []
    for graph in (False, True):
        net = create_hg(stacks, 1, 16, chan, default_batch=B); net.reset_parameters(seed=3); net.train()
        net.use_graph = graph
        opt = RMSprop(net, lr=2.5e-4)
        aug = Augmenter(seed=9)
        for i in range(4):
            loss, pckh, pckh_o = train_step(net, opt, aug, batches[i % 2])
        torch.cuda.synchronize()
        finals.append((net.flat_params.clone(), net.flat_buffers.clone(), net.flat_grads.clone(), float(loss), float(pckh_o)))
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])
    assert abs(finals[0][3] - finals[1][3]) < 1e-6 * abs(finals[0][3]) and finals[0][4] == finals[1][4]


@pytest.mark.parametrize('stacks,chan,B', [(1, 128, 2), (2, 256, 24)])
def test_finalize_in_the_consumer_prologue_gives_the_same_bits(stacks, chan, B):
    """BatchNorm finalize of a residual block's inner tensors (models/asn_stacked_hg.py:19,22) folded into the prologue of the
    consuming convolution at the low-resolution levels (csrc/bn_fin.h, pa_net_set_fin_prologue) against a finalize launch in front
    of every consumer: ONE summation order in both forms, so parameters, running statistics and gradients after 3 steps are BITWISE
    equal -- at the small test size (where even the 64 x 64 level has <= 128 partial rows) and at the benchmark's size."""
    from pose_adv_aug_amd import _lib
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd.stack_hg import train_step
    batches = [DeviceBatch.synthetic(B, seed=60 + k) for k in range(2)]
    finals = []
    for rows in (0, 128):
        net = create_hg(stacks, 1, 16, chan, default_batch=B); net.reset_parameters(seed=5); net.train()
        _lib.check(_lib.lib().pa_net_set_fin_prologue(net._net(B), rows))
        opt = RMSprop(net, lr=2.5e-4)
        aug = Augmenter(seed=11)
        for i in range(3):
            loss, pckh, pckh_o = train_step(net, opt, aug, batches[i % 2])
        torch.cuda.synchronize()
        finals.append((net.flat_params.clone(), net.flat_buffers.clone(), net.flat_grads.clone(), float(loss)))
        del net, opt
    assert torch.isfinite(finals[0][2]).all()
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])
    assert abs(finals[0][3] - finals[1][3]) < 1e-6 * abs(finals[0][3])        # (the reported loss is summed with one float atomic per workgroup)


@pytest.mark.parametrize('stacks,chan,B', [(1, 128, 4), (2, 256, 24)])
def test_meters_beside_the_backward_pass_give_the_same_values(stacks, chan, B):
    """stack-hg.py:171-180's accuracies launched between the forward and the backward pass on the engine's meter stream
    (pa_net_meters_async, stack_hg.train_step) against the reference's order (after the optimizer step, on the net's stream): the same
    loss / accuracy values at every step and bitwise the same parameters, statistics and gradients over 4 steps -- the meters only read
    the forward pass's heat maps, and the net's stream joins them at the end of the backward pass."""
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd.stack_hg import train_step, PCK_IDX
    batches = [DeviceBatch.synthetic(B, seed=70 + k) for k in range(2)]
    finals, meters = [], []
    for beside in (True, False):
        net = create_hg(stacks, 1, 16, chan, default_batch=B); net.reset_parameters(seed=6); net.train()
        opt = RMSprop(net, lr=2.5e-4)
        aug = Augmenter(seed=13)
        vals = []
        for i in range(4):
            if beside:
                loss, pckh, pckh_o = train_step(net, opt, aug, batches[i % 2])
            else:                       # the reference's order, every call on the net's stream
                data = aug.regular(batches[i % 2])
                loss, _ = net.loss_and_backward(img4=data['img4'], pts=data['pts'])
                opt.step()
                pckh = net.accuracy(PCK_IDX)[0]
                pckh_o = net.pckh_origin_res(data['c'], data['s'], data['r'], data['grnd_pts'], data['normalizer'])[0][0]
            junk = torch.empty(1 << 22, device='cuda').normal_()       # allocator traffic between the steps: a freed meter buffer would be reused here
            vals.append((loss, pckh, pckh_o)); del junk
        torch.cuda.synchronize()
        meters.append([(float(a), float(b), float(c)) for a, b, c in vals])
        finals.append((net.flat_params.clone(), net.flat_buffers.clone(), net.flat_grads.clone()))
        del net, opt
    for (l0, a0, o0), (l1, a1, o1) in zip(meters[0], meters[1]):
        assert abs(l0 - l1) < 1e-6 * abs(l1) and a0 == a1 and o0 == o1, (meters[0], meters[1])      # (the reported loss is summed with one float atomic per workgroup)
    assert 0.0 <= meters[0][-1][1] <= 1.0 and 0.0 <= meters[0][-1][2] <= 1.0
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])


def _blob_dataset(n, res, seed):
    """learnable synthetic people: every joint is a colour-coded Gaussian blob in the image at its location"""
    g = inputs.rng(seed, 7)
    hr = res // 4
    pts = g.uniform(6.0, hr - 6.0, size=(n, 16, 2))
    pts[g.random((n, 16)) < 0.1] = 0.0
    yy, xx = np.meshgrid(np.arange(res, dtype=np.float32), np.arange(res, dtype=np.float32), indexing='ij')
    code = np.stack([[(j % 3 == c) * 1.0 + 0.15 * ((j // 3) % 5) for c in range(3)] for j in range(16)]).astype(np.float32)    # [16][3]
    img = np.zeros((n, 3, res, res), dtype=np.float32)
    for i in range(n):
        for j in range(16):
            if pts[i, j, 0] > 0:
                blob = np.exp(-((xx - 4 * pts[i, j, 0]) ** 2 + (yy - 4 * pts[i, j, 1]) ** 2) / (2 * (3.0 + 0.5 * (j // 3)) ** 2))
                img[i] += code[j][:, None, None] * blob[None]
    return np.clip(img, 0, 1), pts


@pytest.mark.parametrize('wseed,dseed', [(23, 5), (29, 6), (31, 7)])
def test_training_trajectory_tracks_the_oracle_over_150_steps(wseed, dseed):
    """The "PCKh-matching" half of the metric at TRAINING level: a 1-stack hourglass (chan 128, B = 4, 128x128) trained for
    150 RMSprop steps on a small learnable set (colour-coded blobs at the joints), engine (bf16 storage) and fp32 oracle
    from the same weights on the same batches.  Both learn: loss falls by > 3x and PCKh@0.5 in heat-map space
    (Evaluation.accuracy, computed by the ORACLE code on each side's own heat maps of HELD-OUT images) rises; at the end
    the two agree on each of three seeds (weights, data) at the bars of round 4: smoothed loss within 5 %, held-out PCKh within
    +-0.08, held-out mse within 12 %.

    Round 5 had widened these bars (9 %, -0.08 / +0.17, 17 %) after seeing three seeds.  Round 6 narrows the STATISTIC instead: the
    held-out set is 20 images (~290 visible joints, one joint = 0.0035) instead of 4 (~58 joints, one joint = 0.017), so that the
    sampling noise of the held-out numbers no longer dominates what two equally good models differ by.

    A third trajectory runs beside the two: the oracle's modules evaluated with the engine's bf16 STORAGE points and bf16 gradient
    roundings (tests/bf16_emul.py), i.e. "the engine's numerics without the engine's kernels".  It is NOT a tighter reference for a
    150-step trajectory -- measured on the three seeds (round 6): emulation vs fp32 oracle tail loss 2.4 - 6.1 %, engine vs fp32 oracle
    0.2 - 0.9 %, engine vs emulation 2.9 - 4.9 %: the three trajectories are mutually as far apart as bf16 storage noise amplified over
    150 steps makes ANY two of them; the emulation-level bars that ARE tight are the one-step ones (first-step loss 3e-3 here,
    test_residual_block / test_gpu_local.py 3e-3 ... 4e-2).  The three-way distances are printed."""
    import copy
    from tests import bf16_emul
    from pose_adv_aug_amd.utils.optim import RMSprop
    torch.set_num_threads(16)
    B, res, chan, steps = 4, 128, 128, 150
    ref, net = _hg_pair(1, chan, B, res, seed=wseed)
    emu = copy.deepcopy(ref)                  # the SAME oracle modules evaluated with the engine's bf16 storage points (tests/bf16_emul.py)
    imgs, pts = _blob_dataset(40, res, seed=dseed)               # 20 training images (5 batches), 20 held out
    heat = inputs.heatmaps_from_pts(pts, res=res // 4)
    opt_ref = ostep.make_optimizer(ref)
    opt_emu = ostep.make_optimizer(emu)
    opt = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8)
    ref.train(); net.train(); emu.train()
    l_ref, l_dev, l_emu = [], [], []
    rg = bf16_emul.ROUND_GRADS
    bf16_emul.ROUND_GRADS = True              # ... and its bf16 gradient storage
    try:
        for i in range(steps):
            sl = slice((i % 5) * B, (i % 5) * B + B)
            _, l = ostep.pose_loss_and_grads(ref, t(imgs[sl]), t(heat[sl])); opt_ref.step(); l_ref.append(float(l))
            le = opl.stack_mse(bf16_emul.emul_hourglass_net(emu, t(imgs[sl])), t(heat[sl]))
            emu.zero_grad(); le.backward(); opt_emu.step(); l_emu.append(float(le))
            l2, _ = net.loss_and_backward(t(imgs[sl]).cuda(), t(pts[sl]).cuda()); opt.step(); l_dev.append(float(l2))
    finally:
        bf16_emul.ROUND_GRADS = rg
    assert abs(l_dev[0] - l_ref[0]) / l_ref[0] < 1e-2
    assert abs(l_dev[0] - l_emu[0]) / l_emu[0] < 3e-3            # (one step: the emulation's distance, not the fp32 oracle's)
    tail = lambda v: float(np.mean(v[-20:]))
    assert tail(l_ref) < l_ref[0] / 3 and tail(l_dev) < l_dev[0] / 3, (l_ref[0], tail(l_ref), l_dev[0], tail(l_dev))
    assert abs(tail(l_dev) - tail(l_ref)) / tail(l_ref) < 0.05, (tail(l_dev), tail(l_ref))
    # held-out images, eval mode (running statistics of 150 steps), PCKh by the oracle's Evaluation on each side's maps
    ref.eval(); net.eval(); emu.eval()
    o_ref, o_dev, o_emu = [], [], []
    for k in range(5, 10):
        sl = slice(k * B, k * B + B)
        with torch.no_grad():
            o_ref.append(ref(t(imgs[sl]))[-1])
            o_emu.append(bf16_emul.emul_hourglass_net(emu, t(imgs[sl]))[-1])
        o_dev.append(net(t(imgs[sl]).cuda(), pts=t(pts[sl]).cuda())[-1].cpu())
        if k == 9:          # ... and the engine's own metric kernel on its own maps says the same as the oracle code on those maps
            assert abs(float(net.accuracy(list(range(16)))[0]) - float(opl.accuracy(o_dev[-1], t(heat[sl]), list(range(16)))[0])) < 1e-4
    o_ref, o_dev, o_emu = torch.cat(o_ref), torch.cat(o_dev), torch.cat(o_emu)
    held = t(heat[5 * B:10 * B])
    idx = list(range(16))
    a_ref, a_dev, a_emu = (float(opl.accuracy(o, held, idx)[0]) for o in (o_ref, o_dev, o_emu))
    v_ref, v_dev, v_emu = (float(((o - held) ** 2).mean()) for o in (o_ref, o_dev, o_emu))
    print('TRAJECTORY observed: tail loss dev %.6g ref %.6g (rel %.4f); held-out PCKh dev %.4f ref %.4f; held-out mse dev %.6g ref %.6g (rel %.4f)'
          % (tail(l_dev), tail(l_ref), abs(tail(l_dev) - tail(l_ref)) / tail(l_ref), a_dev, a_ref, v_dev, v_ref, abs(v_dev - v_ref) / v_ref))
    print('TRAJECTORY vs bf16 emulation: tail loss emu %.6g (rel %.4f); held-out PCKh emu %.4f (dev - emu %+.4f); held-out mse emu %.6g (rel %.4f); '
          'emu vs fp32: tail %.4f PCKh %+.4f mse %.4f'
          % (tail(l_emu), abs(tail(l_dev) - tail(l_emu)) / tail(l_emu), a_emu, a_dev - a_emu, v_emu, abs(v_dev - v_emu) / v_emu,
             abs(tail(l_emu) - tail(l_ref)) / tail(l_ref), a_emu - a_ref, abs(v_emu - v_ref) / v_ref))
    assert a_ref > 0.2 and a_dev > 0.2, (a_ref, a_dev)
    assert abs(a_dev - a_ref) <= 0.08 and abs(v_dev - v_ref) / v_ref < 0.12, (a_ref, a_dev, v_ref, v_dev)


def test_engine_against_the_reference_golden_directly():
    """tests/golden/hg_s1c128.npz holds the outputs of the REFERENCE's own modules (transliterated, make_goldens.py) for a
    width the engine runs: 1-stack, chan 128, B = 2, 128x128, deterministic weights.  The engine is compared with those
    arrays directly (no oracle in between): train-mode heat maps (bf16 storage: 15 % of their RMS, as in the oracle tests),
    loss 1 %, heat-map PCK equal, the output layers' gradients 5 % / cosine .998, eval-mode heat maps after the step."""
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'hg_s1c128.npz'))
    ref = om.create_hg(1, 1, 16, 128)                 # (only the carrier of the deterministic weights and the parameter names)
    om.deterministic_fill_(ref, seed=33)
    net = create_hg(1, 1, 16, 128, res=128, default_batch=2)
    net.load_state_dict(ref.state_dict())
    assert net.num_params() == int(g['nparams'])
    img = t(inputs.images(133, 2, 128)); pts = inputs.heat_pts(233, 2, res=32)
    heat = t(inputs.heatmaps_from_pts(pts, res=32))
    net.train()
    opt = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8)
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    assert abs(float(loss) - float(g['loss'])) / float(g['loss']) < 1e-2
    assert rel_rms(outs[0].cpu(), t(g['out'][0])) < 0.15
    acc = net.accuracy([0, 1, 2, 3, 4, 5, 10, 11, 14, 15]).cpu().numpy()
    assert np.allclose(acc, opl.accuracy(outs[0].cpu(), heat, [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]).numpy(), atol=1e-4)
    grads = dict(net.named_grads())
    mine = torch.cat([grads[str(n)].flatten().cpu() for n in g['head_names']])
    assert rel_rms(mine, t(g['head_grads'])) < 5e-2 and cosine(mine, t(g['head_grads'])) > 0.998
    opt.step()
    net.eval()
    oe = net(img.cuda())
    assert rel_rms(oe[0].cpu(), t(g['out_eval'][0])) < 0.15
