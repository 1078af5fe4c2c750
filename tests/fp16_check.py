"""Parity of the IEEE-half build (libposeadv_hip_fp16.so, BASELINE configs[4]: 8-stack 384x384, "fp16 MFMA 1x1 convs") against
the fp32 CPU oracle.  One storage type per process, so tests/test_gpu_fp16.py runs this file in a child process with
POSEADV_DTYPE=fp16; every check prints a line and the script exits non-zero on the first failure.

Tolerances: half keeps 11 significand bits (rel. step 2^-11 = 4.9e-4, 8x finer than bf16), so single operators agree to
~1e-3; through ~60-200 layers of an untrained, chaotic net the end-to-end bars are those of the bf16 tests."""
import os
import sys

os.environ['POSEADV_DTYPE'] = 'fp16'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import model as om, pylib as opl, step as ostep  # noqa: E402
from tests import inputs  # noqa: E402
from tests.test_gpu_net import rel_rms, cosine, t, _hg_pair  # noqa: E402


def ok(name, cond, detail=''):
    print(('PASS ' if cond else 'FAIL ') + name + ' ' + str(detail), flush=True)
    if not cond:
        sys.exit(1)


def main():
    import pose_adv_aug_amd as P
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    from pose_adv_aug_amd.utils.optim import RMSprop
    torch.set_num_threads(8)
    S = P.grad_scale()
    ok('library is the fp16 build', lib().pa_dtype() == 1 and P.act_dtype() == torch.float16 and S == 32768.0, S)

    # ---- single convolutions through the MFMA kernels (v_mfma_f32_16x16x32_f16), operands rounded to half
    h = lambda x: x.half().float()
    for (B, Cin, Cout, H, k) in [(2, 256, 128, 64, 1), (2, 128, 256, 64, 1), (2, 128, 128, 64, 3), (3, 128, 128, 8, 3), (2, 256, 128, 16, 1)]:
        g = inputs.rng(300 + Cin + k)
        x = t(g.standard_normal((B, Cin, H, H)).astype(np.float32)); w = t((g.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
        bias = t(g.standard_normal(Cout).astype(np.float32)); dy = t(g.standard_normal((B, Cout, H, H)).astype(np.float32))
        ws = torch.zeros(lib().pa_conv2d_workspace_bytes(B, Cin, Cout, H, H, k), dtype=torch.uint8, device='cuda')
        y = torch.empty((B, Cout, H, H), device='cuda'); dx = torch.empty((B, Cin, H, H), device='cuda')
        dw = torch.empty((Cout, Cin, k, k), device='cuda'); db = torch.empty(Cout, device='cuda')
        xc, wc, bc, dyc = x.cuda(), w.cuda(), bias.cuda(), dy.cuda()
        check(lib().pa_conv2d(0, ptr(xc), None, ptr(wc), ptr(bc), ptr(y), None, B, Cin, Cout, H, H, k, ptr(ws), stream()))
        check(lib().pa_conv2d(1, ptr(dyc), None, ptr(wc), None, ptr(dx), None, B, Cin, Cout, H, H, k, ptr(ws), stream()))
        check(lib().pa_conv2d(2, ptr(dyc), ptr(xc), ptr(wc), None, ptr(dw), ptr(db), B, Cin, Cout, H, H, k, ptr(ws), stream()))
        e0 = rel_rms(y.cpu(), F.conv2d(h(x), h(w), bias, padding=k // 2))
        e1 = rel_rms(dx.cpu(), F.conv_transpose2d(h(dy), h(w), padding=k // 2))
        xr = h(x).requires_grad_(False); wr = h(w).clone().requires_grad_(True)
        F.conv2d(xr, wr, None, padding=k // 2).backward(h(dy))
        e2 = rel_rms(dw.cpu(), wr.grad)
        ok('conv %dx%d %d->%d @%d fwd/dgrad/wgrad' % (k, k, Cin, Cout, H), e0 < 1.5e-3 and e1 < 1.5e-3 and e2 < 1.5e-3, (e0, e1, e2))

    # ---- residual block, forward + backward with train-mode BatchNorm
    for (C_, H, B) in [(256, 16, 2), (128, 64, 2)]:
        blk = om.Residual(C_, C_); om.deterministic_fill_(blk, seed=3); blk.train()
        x = torch.relu(t(inputs.rng(4).standard_normal((B, C_, H, H)).astype(np.float32)))
        dy = t(inputs.rng(5).standard_normal((B, C_, H, H)).astype(np.float32))
        xr = h(x).requires_grad_(True)
        y = blk(xr); y.backward(h(dy))
        blk0 = om.Residual(C_, C_); om.deterministic_fill_(blk0, seed=3)                  # running statistics before the step
        bufs = torch.cat([b.flatten().float() for n, b in blk0.named_buffers() if 'num_batches' not in n]).cuda()
        ws = torch.zeros(lib().pa_residual_workspace_bytes(B, H, H, C_), dtype=torch.uint8, device='cuda')
        yo = torch.empty((B, C_, H, H), device='cuda'); dxo = torch.empty_like(yo)
        # the flat parameter layout of the op is the block's own state_dict order with 16-byte aligned tensors
        off, chunks = 0, []
        for p in blk.parameters():
            pad = (-off) % 4
            chunks.append(torch.zeros(pad)); off += pad
            chunks.append(p.detach().flatten()); off += p.numel()
        chunks.append(torch.zeros((-off) % 4))
        params = torch.cat(chunks).cuda(); grads = torch.zeros_like(params)
        xd, dyd = x.cuda(), dy.cuda()          # (kept alive across the call)
        check(lib().pa_residual_fwd_bwd(ptr(xd), ptr(dyd), ptr(params), ptr(yo), ptr(dxo), ptr(grads), ptr(bufs), B, C_, H, H, ptr(ws), stream()))
        e_y, e_dx = rel_rms(yo.cpu(), y.detach()), rel_rms(dxo.cpu(), xr.grad)
        off, worst = 0, 0.0
        for name, p in blk.named_parameters():
            off += (-off) % 4
            gdev = grads[off:off + p.numel()].cpu().view_as(p); off += p.numel()
            if float(p.grad.abs().max()) > 1e-6 and not name.endswith('conv1.bias') and not name.endswith('conv2.bias') and not name.endswith('conv3.bias'):
                worst = max(worst, rel_rms(gdev, p.grad))
        ok('residual block C=%d H=%d' % (C_, H), e_y < 4e-3 and e_dx < 5e-2 and worst < 8e-2, (e_y, e_dx, worst))

    # ---- BASELINE configs[4] shape: 8-stack, 384x384 (96x96 maps), here B = 2
    B, res, chan = 2, 384, 256
    ref, net = _hg_pair(8, chan, B, res, seed=5)
    img = t(inputs.images(31, B, res)); pts = inputs.heat_pts(32, B, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    ref.train(); net.train()
    out_ref, loss_ref = ostep.pose_loss_and_grads(ref, img, heat)
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    ok('8-stack 384: loss vs oracle', abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-2, (float(loss), float(loss_ref)))
    ok('8-stack 384: first-stack heat maps', rel_rms(outs[0].cpu(), out_ref[0].detach()) < 0.1, rel_rms(outs[0].cpu(), out_ref[0].detach()))
    g = net.flat_grads
    ok('8-stack 384: gradients finite and non-zero', bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0)
    gref = dict(ref.named_parameters())
    for name, gd in net.named_grads():
        if name.startswith('out_conv.7.') or name.startswith('linear.7.1.'):
            e, c = rel_rms(gd.cpu() / S, gref[name].grad), cosine(gd.cpu(), gref[name].grad)
            ok('8-stack 384: grad %s / scale vs oracle' % name, e < 5e-2 and c > 0.998, (e, c))
    # gradients deep in the net (the stem, ~200 layers below the last loss) must not have been flushed to zero by half's range
    stem = dict(net.named_grads())['conv1.weight'].cpu() / S
    rs = gref['conv1.weight'].grad
    ok('8-stack 384: stem gradient magnitude survives (no underflow)', 0.5 < float(stem.norm() / rs.norm()) < 2.0 and float((stem == 0).float().mean()) < 0.01,
       (float(stem.norm() / rs.norm()), float((stem == 0).float().mean())))

    # ---- BASELINE configs[4] at ITS OWN size (8-stack, 384x384, B = 16 per GPU): the fp32 oracle takes minutes there, so the checks are
    # the size-independent properties -- the loss the engine reports IS sum_stacks mean((out - gaussian(pts))^2) of the heat maps it
    # returns, every gradient is finite, none of the 8 stacks' or the stem's gradients vanished in half's range, an optimizer step is
    # applied (not skipped) and the next step's loss is finite and lower on the same batch
    B16 = 16
    del net, ref
    torch.cuda.empty_cache()
    _, net = _hg_pair(8, chan, B16, res, seed=5)
    img = t(inputs.images(33, B16, res)); pts = inputs.heat_pts(34, B16, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    net.train()
    opt16 = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8)
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    again = sum(float(((o.cpu() - heat) ** 2).mean()) for o in outs)
    ok('8-stack 384 B=16: reported loss == loss recomputed from the returned heat maps', abs(float(loss) - again) / again < 1e-3, (float(loss), again))
    g = net.flat_grads
    ok('8-stack 384 B=16: gradients finite', bool(torch.isfinite(g).all()))
    gd = dict(net.named_grads())
    dead = [n for n in ['conv1.weight'] + ['hg.%d.skip1.0.conv2.weight' % i for i in range(8)] + ['out_conv.%d.weight' % i for i in range(8)]
            if float(gd[n].abs().max()) == 0.0 or float((gd[n] == 0).float().mean()) > 0.05]
    ok('8-stack 384 B=16: no stack lost its gradient to underflow', not dead, dead)
    before = net.flat_params.clone(); sk0 = opt16.skipped_steps()
    opt16.step()
    loss2, _ = net.loss_and_backward(img.cuda(), t(pts).cuda())
    ok('8-stack 384 B=16: the step is applied and the loss falls', opt16.skipped_steps() == sk0 and not torch.equal(net.flat_params, before)
       and bool(torch.isfinite(net.flat_params).all()) and np.isfinite(float(loss2)) and float(loss2) < float(loss), (float(loss), float(loss2)))
    del net, opt16
    torch.cuda.empty_cache()

    # ---- a few RMSprop steps: the scaled gradients give the oracle's trajectory
    B, res, chan = 2, 128, 128
    ref, net = _hg_pair(1, chan, B, res, seed=17)
    img = t(inputs.images(18, B, res)); pts = inputs.heat_pts(19, B, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    opt_ref, opt = ostep.make_optimizer(ref), RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8)
    ref.train(); net.train()
    lr_, ld_ = [], []
    for _ in range(5):
        _, l = ostep.pose_loss_and_grads(ref, img, heat); opt_ref.step(); lr_.append(float(l))
        l2, _ = net.loss_and_backward(img.cuda(), t(pts).cuda()); opt.step(); ld_.append(float(l2))
    ok('training steps track the oracle', abs(ld_[0] - lr_[0]) / lr_[0] < 1e-2 and ld_[-1] < ld_[0] and all(abs(a - b) / b < 0.25 for a, b in zip(ld_, lr_)), (ld_, lr_))
    # ---- an overflowed backward pass (inf / NaN in the scaled gradient) must not poison the master weights: step skipped + counted
    before_p, before_v, skipped0 = net.flat_params.clone(), opt.square_avg.clone(), opt.skipped_steps()
    net.flat_grads[12345] = float('inf'); net.flat_grads[777] = float('nan')
    opt.step()
    ok('non-finite gradient: step skipped and counted', torch.equal(net.flat_params, before_p) and torch.equal(opt.square_avg, before_v)
       and opt.skipped_steps() == skipped0 + 1, (opt.skipped_steps(), skipped0))
    l3, _ = net.loss_and_backward(img.cuda(), t(pts).cuda()); opt.step()
    ok('the next finite step is applied', not torch.equal(net.flat_params, before_p) and opt.skipped_steps() == skipped0 + 1 and bool(torch.isfinite(net.flat_params).all()))
    print('ALL FP16 CHECKS PASSED', flush=True)


if __name__ == '__main__':
    main()
