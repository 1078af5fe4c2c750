"""GPU parity of the ASN scale/rotation agent and of the joint-training steps (reference
models/asn_stacked_hg.py:349-439, joint-train-pose-s-r-agent.py:195-467)."""
import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import pylib as opl
from oracle import step as ostep
from tests import inputs, bf16_emul
from tests.test_gpu_net import rel_rms, cosine, t

pytestmark = pytest.mark.gpu


def _pair(chan, B, res, seed):
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg, create_asn
    ref = om.create_hg(2, 1, 16, chan); om.deterministic_fill_(ref, seed=seed)
    ragent = om.create_asn(chan, chan, 7, 7, is_aug=True); om.deterministic_fill_(ragent, seed=seed + 1)
    net = create_hg(2, 1, 16, chan, res=res, default_batch=B); net.load_state_dict(ref.state_dict())
    agent = create_asn(chan, chan, 7, 7, is_aug=True, res=res, default_batch=B)
    assert list(agent.state_dict().keys()) == list(ragent.state_dict().keys())
    for (k, a), (_, b) in zip(agent.state_dict().items(), ragent.state_dict().items()):
        assert tuple(a.shape) == tuple(b.shape), k
    agent.load_state_dict(ragent.state_dict())
    return ref, ragent, net, agent


def test_agent_graph_matches_locally():
    """Half-hourglass (eval) + agent (train), node by node at the engine's own operating point (the method
    of tests/test_gpu_local.py): every residual block of the agent, the pool+add merges, the average-pool /
    Linear head, the KL loss and every parameter gradient."""
    import ctypes as C
    import torch.nn.functional as F
    from pose_adv_aug_amd._lib import lib, check, ptr
    from tests.test_gpu_local import _close, _leaf, FWD_TOL, GRAD_TOL, GRAD_COS
    torch.set_num_threads(8)
    bf16_emul.ROUND_GRADS = True        # the engine stages gradient operands in bf16 (standard mixed precision)
    B, res, chan = 4, 256, 128
    ref, ragent, net, agent = _pair(chan, B, res, seed=41)
    img = t(inputs.images(141, B, res))
    net.eval(); agent.train(); ref.eval(); ragent.train()
    ls, lr = net(img.cuda(), asn=agent, is_half_hg=True, is_aug=True)
    hp, ha = net._net(B), agent._net(B)

    def get(fn, h, name, grad=0):
        shp = (C.c_int * 4)()
        check(fn(h, name.encode(), grad, None, shp))
        out = torch.empty(tuple(shp), device='cuda')
        check(fn(h, name.encode(), grad, ptr(out), shp))
        return out.cpu()
    L = lib()
    feats = [get(L.pa_hg_debug_tensor, hp, n) for n in ('hg0.skip1', 'hg0.skip2', 'hg0.skip3', 'hg0.skip4', 'hg0.neck')]
    act = lambda n: get(L.pa_asn_debug_tensor, ha, n, 0)
    grd = lambda n: get(L.pa_asn_debug_tensor, ha, n, 1)
    # coarse end-to-end sanity against the fp32 oracle (chaotic, see test_gpu_local): features within 15 %
    with torch.no_grad():
        f_ref, _, _ = ref.hg[0].agent_features(ref.stem(img))
    assert rel_rms(feats[0], f_ref['skip1']) < 0.15
    errs = []
    # ---- forward, node by node
    in_blocks = [ragent.residual_skip1, ragent.residual_skip2, ragent.residual_skip3, ragent.residual_skip4, ragent.residual_neck]
    for k in range(5):
        _close(errs, 'fwd in%d' % k, act('in%d' % k), bf16_emul.emul_residual(in_blocks[k], feats[k]).detach(), FWD_TOL)
    merges = [ragent.merge1, ragent.merge2, ragent.merge3, ragent.merge4]
    for k in range(4):
        hi = act('in0') if k == 0 else act('merge%d' % (k - 1))
        pa = bf16_emul.R(F.max_pool2d(hi, 2, 2) + act('in%d' % (k + 1)))
        _close(errs, 'fwd pa%d' % k, act('pa%d' % k), pa, FWD_TOL)
        _close(errs, 'fwd merge%d' % k, act('merge%d' % k), bf16_emul.emul_residual(merges[k], act('pa%d' % k)).detach(), FWD_TOL)
    for k in range(3):
        src = act('merge3') if k == 0 else act('deep%d' % (k - 1))
        _close(errs, 'fwd deep%d' % k, act('deep%d' % k), bf16_emul.emul_residual(ragent.deep_merge[k], src).detach(), FWD_TOL)
    top = _leaf(act('deep2'))
    x = F.avg_pool2d(top, 4).flatten(1)
    es, er = ragent.fc_scale(x), ragent.fc_rotation(x)
    _close(errs, 'logits scale', ls.cpu(), es.detach(), 2e-3)
    _close(errs, 'logits rotation', lr.cpu(), er.detach(), 2e-3)
    # ---- loss + backward
    g = inputs.rng(43)
    ps, pr = torch.softmax(ls.cpu(), 1), torch.softmax(lr.cpu(), 1)
    gs = opl.gen_groundtruth(ps, t(g.integers(0, 7, (B, 1))), t(g.random(B).astype(np.float32)), t(g.random(B).astype(np.float32)))
    gr = opl.gen_groundtruth(pr, t(g.integers(0, 7, (B, 1))), t(g.random(B).astype(np.float32)), t(g.random(B).astype(np.float32)))
    loss = agent.loss_and_backward(gs, gr)
    hip_grads = {n: gg.cpu() for n, gg in agent.named_grads()}
    loss_ref = ostep.agent_kl_loss(es, er, gs, gr)
    ragent.zero_grad(); loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) / max(1e-6, abs(float(loss_ref))) < 2e-3, (float(loss), float(loss_ref))
    for n in ('fc_scale.weight', 'fc_scale.bias', 'fc_rotation.weight', 'fc_rotation.bias'):
        _close(errs, 'grad ' + n, hip_grads[n], dict(ragent.named_parameters())[n].grad, 1e-2, 0.9999)
    _close(errs, 'node-grad deep2', grd('deep2'), top.grad * (top.detach() > 0).float(), GRAD_TOL, GRAD_COS)

    def block_bwd(blk, prefix, a_in, out_name, conv1_tol=(GRAD_TOL, GRAD_COS)):
        a = _leaf(a_in)
        a3 = bf16_emul.emul_residual(blk, a)
        blk.zero_grad()
        a3.backward(grd(out_name))
        for n, p in blk.named_parameters():
            full = prefix + n
            if n.endswith('.bias') and 'bn' not in n and float(hip_grads[full].abs().max()) == 0.0:
                continue
            tol, cs = conv1_tol if n == 'conv1.weight' else (GRAD_TOL, GRAD_COS)
            if n.startswith('bn') and n.endswith('.bias'):
                # d(beta) = sum of the masked bf16 gradient over B*H*W values of both signs (cancellation): the whole-block
                # emulation re-derives the ReLU masks from ITS rounding of x1/x2, 4.0 % was measured, cosine stays 0.999
                tol = 6e-2
            _close(errs, 'grad ' + full, hip_grads[full], p.grad, tol, cs)
        return a.grad
    c = block_bwd(ragent.deep_merge[2], 'deep_merge.2.', act('deep1'), 'deep2')
    _close(errs, 'node-grad deep1', grd('deep1'), c * (act('deep1') > 0).float(), GRAD_TOL, GRAD_COS)
    c = block_bwd(ragent.deep_merge[1], 'deep_merge.1.', act('deep0'), 'deep1')
    _close(errs, 'node-grad deep0', grd('deep0'), c * (act('deep0') > 0).float(), GRAD_TOL, GRAD_COS)
    c = block_bwd(ragent.deep_merge[0], 'deep_merge.0.', act('merge3'), 'deep0')
    _close(errs, 'node-grad merge3', grd('merge3'), c * (act('merge3') > 0).float(), GRAD_TOL, GRAD_COS)
    for k in (3, 2, 1, 0):
        c = block_bwd(merges[k], 'merge%d.' % (k + 1), act('pa%d' % k), 'merge%d' % k)
        _close(errs, 'node-grad pa%d' % k, grd('pa%d' % k), c, GRAD_TOL, GRAD_COS)
        hi_name = 'in0' if k == 0 else 'merge%d' % (k - 1)
        hi, lo = _leaf(act(hi_name)), _leaf(act('in%d' % (k + 1)))
        (F.max_pool2d(hi, 2, 2) + lo).backward(grd('pa%d' % k))
        _close(errs, 'node-grad ' + hi_name, grd(hi_name), hi.grad * (hi.detach() > 0).float(), GRAD_TOL, GRAD_COS)
        _close(errs, 'node-grad in%d' % (k + 1), grd('in%d' % (k + 1)), lo.grad * (lo.detach() > 0).float(), GRAD_TOL, GRAD_COS)
    names = ['residual_skip1.', 'residual_skip2.', 'residual_skip3.', 'residual_skip4.', 'residual_neck.']
    # conv1 of the five input blocks: dW = sum_m g[m] * a[m] with g zero-mean per channel (BatchNorm backward) and
    # `a` the pose net's all-positive features with a large mean -> the sum cancels almost completely, and the
    # bf16 rounding of g (its instance differs between engine and emulation) is amplified by mean(a)/std(a)/corr.
    # Ill-conditioned in ANY bf16 mixed-precision implementation; bounded here at 25 % / cosine 0.97.
    for k in range(5):
        block_bwd(in_blocks[k], names[k], feats[k], 'in%d' % k, conv1_tol=(0.25, 0.97))
    assert not errs, '%d local mismatches, first: %s' % (len(errs), errs[:12])
    assert float(net.flat_grads.abs().max()) == 0.0          # the features are detached: nothing reaches the pose net


def test_joint_training_steps_run_and_learn():
    """train_hg (regular / agent-augmented alternation) and train_agent_sr at the BASELINE batch size, finite and sane."""
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg, create_asn
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd import joint_train_pose_s_r_agent as J
    B = 24                                                      # BASELINE configs[3]: bs 24 per GPU
    hg = create_hg(2, 1, 16, 256, default_batch=B); hg.reset_parameters(seed=1)
    agent = create_asn(256, 256, 7, 7, is_aug=True, default_batch=B); agent.reset_parameters(seed=2)
    assert agent.num_params() == 2577934 and hg.num_params() == 6570784
    opt_hg, opt_sr = RMSprop(hg, lr=2.5e-4), RMSprop(agent, lr=5e-5)
    aug = Augmenter(seed=3)
    batch = DeviceBatch.synthetic(B, seed=4)
    kinds, losses = [], []
    for i in range(4):
        kind, loss, pckh = J.train_hg_step(i, hg, opt_hg, agent, aug, batch, seed=0)
        kinds.append(kind); losses.append(float(loss))
        assert np.isfinite(float(loss)) and 0.0 <= float(pckh) <= 1.0
    assert kinds == ['regular', 'agent', 'regular', 'agent']
    before = agent.flat_params.clone()
    l = J.train_agent_sr(batch, hg, agent, opt_sr, aug, epoch_sr=0, seed=0)
    assert np.isfinite(float(l)) and float(l) >= -1e-6          # a KL divergence
    assert float((agent.flat_params - before).abs().max()) > 0  # the agent moved
    assert bool(torch.isfinite(agent.flat_params).all()) and bool(torch.isfinite(hg.flat_params).all())


def _as_oracle_crops(ds):
    """the engine's crop dictionaries (device) -> what the reference's loader yields (CPU tensors + Gaussian targets)"""
    out = []
    for d in ds:
        pts = d['pts'].cpu().numpy()
        out.append(dict(img=d['img'].cpu(), heatmap=t(inputs.heatmaps_from_pts(pts, res=64)), c=d['c'].cpu(),
                        s=d['s'].cpu().view(-1, 1), r=d['r'].cpu().view(-1, 1), grnd_pts=d['grnd_pts'].cpu(), normalizer=d['normalizer'].cpu()))
    return out


@pytest.mark.parametrize('override', [False, True])
def test_agent_update_matches_the_oracle_restatement(override):
    """train_agent_sr (joint-train-pose-s-r-agent.py:317-422) as ONE unit against oracle/step.py:train_agent_sr: the engine
    records its crops (device crop, parity-pinned in test_gpu_crop.py), the sampled bins and every intermediate; the oracle
    replays the same batch with the same bins.  Asserted: the softmax outputs (2e-2), the four per-person PCKh vectors
    (computed by each side on ITS OWN heat maps: mean |diff| <= 0.05, one flipped joint of 4 x 13; observed 0 - 0.02), the reward-shaped targets -- engine vs the oracle's
    gen_groundtruth on the engine's own (probabilities, bins, PCKh) to 1e-6 and vs the full oracle to 2e-2 (observed 1.2e-2) --, the KL loss
    (formula: 1e-5 on the engine's operands; end to end: 3.5 %, observed 1.9 %), and the direction of the RMSprop step of the two Linear
    heads (first step = -lr * 10 * sign(g): cosine of the parameter deltas >= 0.9).  override=True feeds both sides the
    same non-trivial PCKh vectors so that BOTH branches of the reward shaping and its clamp run."""
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd import joint_train_pose_s_r_agent as J
    torch.set_num_threads(8)
    B, res, chan = 4, 256, 128
    ref, ragent, net, agent = _pair(chan, B, res, seed=61)
    opt_sr = RMSprop(agent, lr=5e-5)
    opt_ref = ostep.make_optimizer(ragent, lr=5e-5)
    batch = DeviceBatch.synthetic(B, seed=62)
    aug = Augmenter(seed=63)
    ov = None
    if override:
        g = inputs.rng(64)
        ov = ([t((g.integers(0, 15, B) / 14.0).astype(np.float32)) for _ in range(2)], [t((g.integers(0, 15, B) / 14.0).astype(np.float32)) for _ in range(2)])
        ov[1][0][0] = ov[0][0][0]                      # a tie: "not harder" branch
    before = {k: v.clone() for k, v in agent.state_dict().items()}
    rbefore = {k: v.clone() for k, v in ragent.state_dict().items()}
    tr = {}
    loss = J.train_agent_sr(batch, net, agent, opt_sr, aug, epoch_sr=3, seed=5, trace=tr,
                            pckh_override=None if ov is None else ([v.cuda() for v in ov[0]], [v.cuda() for v in ov[1]]))
    si, ri = tr['bins'][0].cpu().long(), tr['bins'][1].cpu().long()
    o = ostep.train_agent_sr(ref, ragent, opt_ref, tr['std']['img'].cpu(), _as_oracle_crops(tr['regular']), _as_oracle_crops(tr['agent']),
                             si, ri, pckh_override=ov)
    for k in range(2):
        assert float((tr['probs'][k].cpu() - o['probs'][k]).abs().max()) < 2e-2
        assert float((tr['pckh_regular'][k].cpu() - o['pckh_regular'][k]).abs().mean()) <= 0.05
        assert float((tr['pckh_agent'][k].cpu() - o['pckh_agent'][k]).abs().mean()) <= 0.05
        mine = opl.gen_groundtruth(tr['probs'][k].cpu(), (si, ri)[k].view(-1, 1), tr['pckh_regular'][k].cpu(), tr['pckh_agent'][k].cpu())
        assert float((tr['targets'][k].cpu() - mine).abs().max()) < 1e-6
        assert abs(float(mine.sum()) - B) < 1e-5
        if override:
            assert float((tr['targets'][k].cpu() - o['targets'][k]).abs().max()) < 2e-2
    l_formula = float(ostep.agent_kl_loss(tr['logits'][0].cpu(), tr['logits'][1].cpu(), tr['targets'][0].cpu(), tr['targets'][1].cpu()))
    assert abs(float(loss) - l_formula) < 1e-5 + 1e-4 * abs(l_formula), (float(loss), l_formula)
    if override:
        assert abs(float(loss) - float(o['loss'])) < 3.5e-2 * abs(float(o['loss'])) + 1e-4, (float(loss), float(o['loss']))
        rsd = ragent.state_dict()
        for name in ('fc_scale.weight', 'fc_rotation.weight'):
            d_dev = (agent.state_dict()[name].cpu() - before[name].cpu()).flatten()
            d_ref = (rsd[name] - rbefore[name]).flatten()
            assert float(d_dev.abs().max()) > 0 and cosine(d_dev, d_ref) >= 0.9, (name, cosine(d_dev, d_ref))
