"""GPU parity of the occlusion (dropout) agent branch, SURVEY.md section 8f rank 4 (reference models/asn_stacked_hg.py:
_dropout :79-100, _sample_mask :102-136, routing :172-190 and :308-324, ASN head :378-379,437-439).  The oracle
(oracle/model.py) is pinned to the transliterated reference by tests/test_oracle_golden.py::test_occlusion_agent_branch."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import step as ostep
from tests import inputs
from tests.test_gpu_net import rel_rms, cosine, t

pytestmark = pytest.mark.gpu


def bf16r(x):
    return x.to(torch.bfloat16).float()


def _dbg(fn, h, name, grad=0):
    from pose_adv_aug_amd._lib import check, ptr
    shp = (C.c_int * 4)()
    check(fn(h, name.encode(), grad, None, shp))
    out = torch.empty(tuple(shp), device='cuda')
    check(fn(h, name.encode(), grad, ptr(out), shp))
    return out.cpu()


def _up(masks, H):
    return torch.nn.functional.interpolate(masks, scale_factor=H // 4, mode='nearest') if H > 4 else masks


def test_cell_mask_operator_is_exact():
    from pose_adv_aug_amd.models.asn_stacked_hg import dropout
    g = inputs.rng(301)
    for (B, Cc, H) in ((2, 8, 4), (3, 16, 16), (2, 256, 64)):
        x = bf16r(t(g.standard_normal((B, Cc, H, H)).astype(np.float32)))
        masks = torch.ones(B, 1, 4, 4)
        for b in range(B):
            for cell in g.choice(16, 2, replace=False):
                masks[b, 0, cell // 4, cell % 4] = 0
        want = om.Hourglass.dropout(x, masks)
        got = dropout(x.cuda(), masks.cuda()).cpu()
        assert torch.equal(got, want), (B, Cc, H)


def test_mask_sampler_law_and_replay():
    """pa_sample_dropout_masks: (a) with given uniforms it is the oracle's sequential inverse-CDF draw, cell for cell;
    (b) with its own stream the pair frequencies follow p_a p_b / (1 - p_a) (np.random.choice(replace=False))."""
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    from pose_adv_aug_amd.models.asn_stacked_hg import sample_mask
    g = inputs.rng(302)
    B = 512
    lg = t(g.normal(0, 2.0, (B, 1, 4, 4)).astype(np.float32)).cuda()
    u = t(g.random((B, 2)))
    probs = torch.empty((B, 16), device='cuda'); masks = torch.empty((B, 16), device='cuda')
    idx = torch.empty((B, 2), dtype=torch.int32, device='cuda')
    uu = u.cuda()
    check(lib().pa_sample_dropout_masks(ptr(lg.reshape(B, 16).contiguous()), B, 16, 2, 0, 0, ptr(uu), ptr(probs), ptr(masks), ptr(idx), stream()))
    want = om.sample_cells_inverse_cdf(probs.cpu().numpy().astype(np.float64), u.numpy())
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.allclose(probs.cpu().numpy(), torch.softmax(lg.cpu().reshape(B, 16), 1).numpy(), atol=1e-6)
    assert torch.equal(masks.cpu().view(B, 1, 4, 4), om.masks_from_indexes(t(want)))
    m2, i2 = sample_mask(lg, uniforms=u)                       # the python wrapper, same draw
    assert torch.equal(i2.cpu(), t(want)) and torch.equal(m2.cpu(), masks.cpu().view(B, 1, 4, 4))
    # (b) own stream: one logit row repeated, 40000 draws
    n = 40000
    row = torch.tensor([1.2, 0.3, -0.5, 0.0, 2.0, -1.0, 0.5, 0.1, -2.0, 0.7, 0.2, -0.3, 1.0, -0.8, 0.4, 0.0])
    p = torch.softmax(row, 0).double().numpy()
    m, i = sample_mask(row.view(1, 1, 4, 4).repeat(n, 1, 1, 1).cuda(), seed=5, step=3)
    i = i.cpu().numpy()
    assert (i[:, 0] != i[:, 1]).all() and float(m.sum()) == n * 14
    m_b, i_b = sample_mask(row.view(1, 1, 4, 4).repeat(n, 1, 1, 1).cuda(), seed=5, step=3)
    assert torch.equal(i_b.cpu(), t(i))                          # counter-based: same (seed, step) -> same draw
    _, i_c = sample_mask(row.view(1, 1, 4, 4).repeat(n, 1, 1, 1).cuda(), seed=5, step=4)
    assert not torch.equal(i_c.cpu(), t(i))
    first = np.bincount(i[:, 0], minlength=16) / n
    assert np.abs(first - p).max() < 0.01
    for a, b in ((4, 0), (0, 4), (12, 9), (4, 12)):
        want_ab = p[a] * p[b] / (1 - p[a])
        assert abs(np.mean((i[:, 0] == a) & (i[:, 1] == b)) - want_ab) < 0.005, (a, b)


def _pair(chan, B, seed):
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg, create_asn
    ref = om.create_hg(2, 1, 16, chan); om.deterministic_fill_(ref, seed=seed)
    ragent = om.create_asn(chan, chan, is_dropout=True); om.deterministic_fill_(ragent, seed=seed + 1)
    net = create_hg(2, 1, 16, chan, res=256, default_batch=B); net.load_state_dict(ref.state_dict())
    agent = create_asn(chan, chan, is_dropout=True, res=256, default_batch=B)
    assert list(agent.state_dict().keys()) == list(ragent.state_dict().keys())
    for (k, a), (_, b) in zip(agent.state_dict().items(), ragent.state_dict().items()):
        assert tuple(a.shape) == tuple(b.shape), k
    agent.load_state_dict(ragent.state_dict())
    return ref, ragent, net, agent


def test_occlusion_branch_forward_backward():
    from pose_adv_aug_amd._lib import lib
    torch.set_num_threads(8)
    B, chan = 2, 128
    ref, ragent, net, agent = _pair(chan, B, seed=61)
    img = t(inputs.images(161, B, 256))
    pts = inputs.heat_pts(162, B, res=64)
    heat = t(inputs.heatmaps_from_pts(pts, res=64))
    L = lib()
    ref.train(); ragent.train(); net.train(); agent.train()
    running0 = net.flat_buffers.clone()
    # ---- half hourglass: mask logits = out_conv on the engine's own deep_merge output (local, tight) ...
    pm_half = net(img.cuda(), asn=agent, is_half_hg=True, is_dropout=True)
    assert tuple(pm_half.shape) == (B, 1, 4, 4)
    hp, ha = net._net(B), agent._net(B)
    top = _dbg(L.pa_asn_debug_tensor, ha, 'deep2')
    want = ragent.out_conv(top).detach()
    assert rel_rms(pm_half.cpu(), want) < 2e-3
    # ... and end to end against the fp32 oracle (the agent reads chaotic features: coarse)
    import copy
    with torch.no_grad():
        pm_ref = copy.deepcopy(ref)(img, copy.deepcopy(ragent), is_half_hg=True, is_dropout=True)
    assert rel_rms(pm_half.cpu(), pm_ref) < 0.25 and cosine(pm_half.cpu() - pm_half.cpu().mean(), pm_ref - pm_ref.mean()) > 0.9
    # ---- whole hourglass: draw, masked forward in BOTH stacks
    net.flat_buffers.copy_(running0); net._nbt = 0
    agent2_nbt = agent._nbt
    outs, pm, idx = net(img.cuda(), asn=agent, is_dropout=True, seed=9)
    masks = net.last_dropout_masks.cpu()
    assert torch.equal(pm.cpu(), pm_half.cpu())                       # same features, same agent, batch statistics
    assert tuple(idx.shape) == (B, 2) and torch.equal(masks, om.masks_from_indexes(idx.cpu()))
    assert net._nbt == 1 and agent._nbt == agent2_nbt + 1              # one running-statistics update each
    for i in (0, 1):
        for k in (1, 2, 3, 4):
            v = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.skip%d' % (i, k))
            mv = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.maskedskip%d' % (i, k))
            assert torch.equal(mv, bf16r(v) * _up(masks, v.shape[2])), (i, k)
        v = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.neck' % i)
        assert torch.equal(_dbg(L.pa_hg_debug_tensor, hp, 'hg%d.maskedneck' % i), bf16r(v) * masks)
        # the decoder consumed the masked tensors: merge4 = up4(masked neck) upsampled + masked skip4
        m4 = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.merge4' % i)
        up4 = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.up4' % i)
        sk4 = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.maskedskip4' % i)
        want4 = torch.nn.functional.interpolate(up4, scale_factor=2, mode='nearest') + sk4
        assert rel_rms(m4, want4) < 4e-3
    # oracle replay of the same draw
    picks = iter(idx.cpu().numpy())
    ref2, ragent2 = copy.deepcopy(ref), copy.deepcopy(ragent)
    out_ref, _, idx_ref = ref2(img, ragent2, is_dropout=True, choice=lambda K, k, p, replace: next(picks))
    assert torch.equal(idx_ref, idx.cpu())
    loss_ref = sum(((o - heat) ** 2).sum() / o.numel() for o in out_ref)
    ref2.zero_grad(); loss_ref.backward()
    with torch.no_grad():
        out_plain = copy.deepcopy(ref)(img)
    # ---- loss + backward through the masks
    net.flat_buffers.copy_(running0)
    plain = net(img.cuda())                                            # (the same batch without the branch)
    net.flat_buffers.copy_(running0)
    loss, outs2 = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True, dropout_masks=masks)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)                                       # same masks -> same forward, bit for bit
    loss_ref = float(loss_ref.detach())
    assert abs(float(loss) - loss_ref) / loss_ref < 1e-2, (float(loss), loss_ref)
    for o, r in zip(outs2, out_ref):
        assert rel_rms(o.cpu(), r.detach()) < 0.15
    # the masks change the outputs the way they change the oracle's
    d_hip, d_ref = (outs2[-1] - plain[-1]).cpu(), (out_ref[-1] - out_plain[-1]).detach()
    assert float(d_ref.abs().max()) > 0 and cosine(d_hip, d_ref) > 0.8, cosine(d_hip, d_ref)
    gref = dict(ref2.named_parameters())
    for name, g in net.named_grads():
        if name.startswith('out_conv.1.') or name.startswith('linear.1.1.'):
            assert rel_rms(g.cpu(), gref[name].grad) < 5e-2 and cosine(g.cpu(), gref[name].grad) > 0.998, name
    # gradients: nothing flows into an occluded cell; elsewhere d(skip) = d(masked skip) * ReLU mask, exactly
    net.flat_buffers.copy_(running0)
    for i in (0, 1):
        for k in (1, 2, 3, 4):
            v = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.skip%d' % (i, k))
            g_m = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.maskedskip%d' % (i, k), 1)
            g_s = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.skip%d' % (i, k), 1)
            assert torch.equal(g_s, g_m * _up(masks, v.shape[2]) * (v > 0).float()), (i, k)
            assert float(g_m.abs().max()) > 0
        v = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.neck' % i)
        g_m = _dbg(L.pa_hg_debug_tensor, hp, 'hg%d.maskedneck' % i, 1)
        assert torch.equal(_dbg(L.pa_hg_debug_tensor, hp, 'hg%d.neck' % i, 1), g_m * masks * (v > 0).float())
    # without masks the branch is off again
    net.flat_buffers.copy_(running0)
    plain2 = net(img.cuda())
    assert torch.equal(plain2[-1], plain[-1])
    # ---- agent backward from a given gradient of the mask logits
    pm3 = net(img.cuda(), asn=agent, is_half_hg=True, is_dropout=True)
    top = _dbg(L.pa_asn_debug_tensor, ha, 'deep2').requires_grad_(True)
    gy = t(inputs.rng(163).standard_normal((B, 1, 4, 4)).astype(np.float32))
    net.zero_grad()
    agent.backward_masks(gy.cuda())
    ragent.zero_grad()
    ragent.out_conv(top).backward(gy)
    hip = {n: g.cpu() for n, g in agent.named_grads()}
    assert rel_rms(hip['out_conv.weight'], ragent.out_conv.weight.grad) < 1e-2
    assert rel_rms(hip['out_conv.bias'], ragent.out_conv.bias.grad) < 1e-4
    g_top = _dbg(L.pa_asn_debug_tensor, ha, 'deep2', 1)
    assert rel_rms(g_top, top.grad * (top.detach() > 0).float()) < 1e-2
    for n, g in hip.items():
        assert bool(torch.isfinite(g).all()), n
        if n.endswith('conv2.weight'):
            assert float(g.abs().max()) > 0, n                         # the trunk received the gradient
    assert float(net.flat_grads.abs().max()) == 0.0                    # detached features: nothing reaches the pose net


def test_occlusion_errors_are_loud():
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg, create_asn
    from pose_adv_aug_amd._lib import lib, PoseAdvError
    with pytest.raises(AssertionError):
        create_asn(128, 128, 7, 7, is_aug=True, is_dropout=True)
    assert not lib().pa_asn_create_dropout(128, 2, 384)                # 6x6 neck map: no 4x4 cell mask
    net = create_hg(1, 1, 16, 128, res=128, default_batch=2)
    with pytest.raises(PoseAdvError):
        net(torch.rand(2, 3, 128, 128).cuda(), dropout_masks=torch.ones(2, 1, 4, 4))
