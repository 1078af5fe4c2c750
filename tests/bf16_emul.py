"""Test infrastructure: PyTorch-CPU emulation of the HIP engine's NUMERICS for one residual block
(reference models/asn_stacked_hg.py:30-49) -- fp32 arithmetic everywhere, but tensors rounded to bf16
at exactly the points where the kernels store or stage them:

  forward : conv outputs (incl. bias and shortcut) stored bf16; BatchNorm statistics taken from the
            stored values; relu(bn(x)) rounded to bf16 when staged as the next MFMA operand
  backward: masked gradients dz stored bf16; the BatchNorm-backward combination kA*dz + kB*x + kC is
            rounded to bf16 when staged as an MFMA operand but kept fp32 when used as an epilogue addend

Comparing the HIP block against this emulation isolates kernel bugs (tight tolerance) from the
unavoidable bf16-vs-fp32 storage difference, which is measured separately against the fp32 oracle."""
import torch
import torch.nn.functional as F


def r16(t):
    return t.bfloat16().float()


def _bn_coeffs(x, gamma, beta, eps=1e-5):
    mean = x.mean(dim=(0, 2, 3))
    var = (x * x).mean(dim=(0, 2, 3)) - mean * mean
    invstd = torch.rsqrt(var.clamp_min(0) + eps)
    s = gamma * invstd
    return s, beta - mean * s, mean, invstd


def _c(v):
    return v.view(1, -1, 1, 1)


def residual_fwd_bwd(blk, x, dy):
    """blk: oracle.model.Residual (identity shortcut); x, dy: fp32 NCHW.  Returns y, dx and a dict of
    parameter gradients keyed like blk.named_parameters()."""
    w1, w2, w3 = r16(blk.conv1.weight.detach()), r16(blk.conv2.weight.detach()), r16(blk.conv3.weight.detach())
    b1, b2, b3 = blk.conv1.bias.detach(), blk.conv2.bias.detach(), blk.conv3.bias.detach()
    a0 = r16(x)
    x1 = r16(F.conv2d(a0, w1, b1))
    s1, t1, m1, i1 = _bn_coeffs(x1, blk.bn1.weight.detach(), blk.bn1.bias.detach())
    a1 = r16(torch.relu(_c(s1) * x1 + _c(t1)))
    x2 = r16(F.conv2d(a1, w2, b2, padding=1))
    s2, t2, m2, i2 = _bn_coeffs(x2, blk.bn2.weight.detach(), blk.bn2.bias.detach())
    a2 = r16(torch.relu(_c(s2) * x2 + _c(t2)))
    x3 = r16(F.conv2d(a2, w3, b3) + a0)
    s3, t3, m3, i3 = _bn_coeffs(x3, blk.bn3.weight.detach(), blk.bn3.bias.detach())
    y = torch.relu(_c(s3) * x3 + _c(t3))

    M = x.shape[0] * x.shape[2] * x.shape[3]
    grads = {}

    def bn_bwd(da, xr, s, t, mean, invstd, name):
        dz = r16(da * ((_c(s) * xr + _c(t)) > 0).float())
        S1 = dz.sum(dim=(0, 2, 3))
        S2 = (dz * (xr - _c(mean)) * _c(invstd)).sum(dim=(0, 2, 3))
        kA = s
        kB = -s * invstd * S2 / M
        kC = -s * S1 / M - kB * mean
        grads[name + '.weight'] = S2
        grads[name + '.bias'] = S1
        return _c(kA) * dz + _c(kB) * xr + _c(kC)          # fp32 gradient w.r.t. the raw conv output

    def wgrad(g16, a16, k):
        # dW[n][c][ky][kx] = sum over pixels of g[n] * a[c] shifted; fp32 accumulation
        return torch.nn.grad.conv2d_weight(a16, (g16.shape[1], a16.shape[1], k, k), g16, padding=k // 2)

    g3f = bn_bwd(r16(dy), x3, s3, t3, m3, i3, 'bn3')
    g3 = r16(g3f)
    grads['conv3.weight'] = wgrad(g3, a2, 1)
    da2 = F.conv_transpose2d(g3, w3)
    g2 = r16(bn_bwd(da2, x2, s2, t2, m2, i2, 'bn2'))
    grads['conv2.weight'] = wgrad(g2, a1, 3)
    da1 = F.conv_transpose2d(g2, w2, padding=1)
    g1 = r16(bn_bwd(da1, x1, s1, t1, m1, i1, 'bn1'))
    grads['conv1.weight'] = wgrad(g1, a0, 1)
    dx = r16(F.conv_transpose2d(g1, w1) + g3f)
    for k in ('conv1', 'conv2', 'conv3'):
        grads[k + '.bias'] = torch.zeros_like(getattr(blk, k).bias)
    return y, dx, grads


# ---------------------------------------------------------------------------------------------
# Whole-network emulation: the oracle's modules (oracle/model.py, same parameters) evaluated with the
# engine's bf16 STORAGE points, rounding being straight-through for autograd.  Forward values then
# carry the same ReLU masks / pool arg-maxes as the HIP path (up to accumulation-order noise), so the
# hand-written backward can be checked against autograd tightly even on an untrained (chaotic) net.
def R(x):
    """bf16 round with identity gradient."""
    return x + (x.bfloat16().float() - x).detach()


class _GradRound(torch.autograd.Function):
    """identity in the forward pass, rounds the GRADIENT to bf16 in the backward pass -- placed where the engine
    stores a gradient tensor in bf16 (the masked gradient dz) or stages one as a bf16 MFMA operand"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


ROUND_GRADS = False          # set True to emulate the engine's bf16 gradient storage as well


def GR(x):
    return _GradRound.apply(x) if ROUND_GRADS else x


def _bn(bn, x):
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)


def _conv(conv, a):
    return F.conv2d(R(a), R(conv.weight), conv.bias, stride=conv.stride, padding=conv.padding)


def emul_residual(blk, a, tap=None, name=''):
    # GR(...) marks the engine's bf16 gradient roundings: the masked gradient in front of every ReLU and the
    # BatchNorm-backward result staged as the MFMA operand of each conv's dgrad / wgrad (the shortcut addend
    # of conv3 stays fp32 in the epilogue)
    x1 = R(GR(_conv(blk.conv1, a)))
    a1 = torch.relu(GR(_bn(blk.bn1, x1)))
    x2 = R(GR(_conv(blk.conv2, a1)))
    a2 = torch.relu(GR(_bn(blk.bn2, x2)))
    sc = a if blk.adapter is None else R(GR(_conv(blk.adapter, a)))
    x3 = R(GR(_conv(blk.conv3, a2)) + sc)
    a3 = torch.relu(GR(_bn(blk.bn3, x3)))
    if tap is not None:
        tap[name + '.x1'] = a1; tap[name + '.x2'] = a2; tap[name] = a3
        for v in (a1, a2, a3):
            if v.requires_grad:
                v.retain_grad()
    return a3


def _rec(tap, name, v):
    if tap is not None:
        tap[name] = v
        if v.requires_grad:
            v.retain_grad()
    return v


def emul_hourglass(hg, a, tap=None, name=''):
    skips = []
    for lvl in (1, 2, 3, 4):
        skips.append(emul_residual(getattr(hg, 'skip%d' % lvl)[0], a, tap, '%s.skip%d' % (name, lvl)))
        a = _rec(tap, '%s.pool%d' % (name, lvl), R(F.max_pool2d(a, 2, 2)))
        a = emul_residual(getattr(hg, 'down%d' % lvl)[0], a, tap, '%s.down%d' % (name, lvl))
    a = emul_residual(hg.neck[0], a, tap, name + '.neck')
    for lvl in (4, 3, 2, 1):
        a = emul_residual(getattr(hg, 'up%d' % lvl)[0], a, tap, '%s.up%d' % (name, lvl))
        a = _rec(tap, '%s.merge%d' % (name, lvl), R(F.interpolate(a, scale_factor=2, mode='nearest') + skips[lvl - 1]))
    return a


def emul_hourglass_net(net, img, tap=None):
    """Names recorded in `tap` match pa_hg_debug_tensor (include/poseadv.h)."""
    a = _rec(tap, 'stem', torch.relu(_bn(net.bn1, R(_conv(net.conv1, img)))))
    a = emul_residual(net.residual1, a, tap, 'res1')
    a = _rec(tap, 'pool0', R(F.max_pool2d(a, 2, 2)))
    a = emul_residual(net.residual2, a, tap, 'res2')
    a = emul_residual(net.residual3, a, tap, 'res3')
    outs = []
    for i in range(net.num_stacks):
        _rec(tap, 'xin%d' % i, a)
        y = emul_hourglass(net.hg[i], a, tap, 'hg%d' % i)
        y = emul_residual(net.post_res[i][0], y, tap, 'post%d' % i)
        lin, lbn = net.linear[i][0], net.linear[i][1]
        l = _rec(tap, 'lin%d' % i, torch.relu(_bn(lbn, R(_conv(lin, y)))))
        heat = _conv(net.out_conv[i], l)                     # kept fp32 by the head kernel
        outs.append(heat)
        if i < net.num_stacks - 1:
            tmp = R(_conv(net.forth_conv[i], l) + a)
            a = R(_conv(net.in_conv[i], heat) + tmp)
    return outs
