"""Validation path (reference stack-hg.py:191-260): eval-mode forward, flip test-time augmentation, PCKh,
final predictions, prediction export.

Bit-exact parts (fp32 adds / integer index work): the W-mirror of the network input, the flip-back +
left/right channel swap + average, and everything downstream of the merged heat maps (arg-max, quarter-pixel
refinement, back-projection, PCKh) -- checked against the CPU oracle applied to the ENGINE's own heat maps.
The heat maps themselves are compared with the fp32 oracle network loosely (bf16 storage, eval-mode
BatchNorm of an untrained net; tolerance 15 % rel-rms as in tests/test_gpu_net.py)."""
import numpy as np
import pytest
import torch

from oracle import pylib as opl
from oracle import step as ostep
from tests import inputs
from tests.test_gpu_net import _hg_pair, rel_rms

pytestmark = pytest.mark.gpu


def test_flip_kernels_are_exact():
    from pose_adv_aug_amd.pylib import HumanAug
    g = inputs.rng(41)
    img4 = torch.from_numpy(g.standard_normal((3, 32, 48, 4)).astype(np.float32)).bfloat16().cuda()
    assert torch.equal(HumanAug.flip_lr_img4(img4).cpu(), img4.cpu().flip(2))
    a = torch.from_numpy(g.standard_normal((3, 16, 20, 24)).astype(np.float32))
    b = torch.from_numpy(g.standard_normal((3, 16, 20, 24)).astype(np.float32))
    got = HumanAug.flip_tta_merge(a.cuda(), b.cuda()).cpu()
    assert torch.equal(got, ostep.validate_tta(a, b))
    # the reference's two in-place helpers give the same thing
    ref = (a + HumanAug.shuffle_channels_for_horizontal_flipping(HumanAug.flip_channels(b.clone()))) / 2
    assert torch.equal(got, ref)


def test_validate_step_matches_oracle_on_the_engines_heat_maps():
    from pose_adv_aug_amd.stack_hg import validate_step, PCK_IDX
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd.pylib import HumanAug
    torch.set_num_threads(8)
    B, res, chan = 4, 256, 128
    ref, net = _hg_pair(2, chan, B, res, seed=11)
    ref.eval(); net.eval()
    batch = DeviceBatch.synthetic(B, seed=3)
    aug = Augmenter(seed=5)
    loss, pckh, pckh_o, preds, output = validate_step(net, aug, batch)
    data = aug.standard(batch)
    # the two forwards separately, merged by the oracle
    o1 = [o.cpu() for o in net.forward(img4=data['img4'], pts=data['pts'])]
    loss1 = float(net._last_losses.sum())
    o2 = [o.cpu() for o in net.forward(img4=HumanAug.flip_lr_img4(data['img4']))]
    merged = ostep.validate_tta(o1[-1], o2[-1])
    assert torch.equal(output.cpu(), merged)
    assert abs(float(loss) - loss1) < 1e-6 * max(1.0, abs(loss1))
    # downstream of the merged heat maps: oracle on the same numbers
    c, s, r = data['c'].cpu(), data['s'].cpu(), data['r'].cpu()
    g, nm = data['grnd_pts'].cpu(), data['normalizer'].cpu()
    heat = torch.from_numpy(inputs.heatmaps_from_pts(data['pts'].cpu().numpy(), res=res // 4))
    want_preds = opl.final_preds(merged, c, s, [res // 4, res // 4], r)
    assert torch.equal(preds.cpu(), want_preds.float())
    assert abs(float(pckh) - float(opl.accuracy(merged, heat, PCK_IDX)[0])) <= 1e-6
    assert abs(float(pckh_o) - float(opl.accuracy_origin_res(merged, c, s, [res // 4, res // 4], g, nm, r)[0])) <= 1e-6
    # the whole thing against the fp32 oracle network on the same warped input
    img = data['img4'][..., :3].float().permute(0, 3, 1, 2).contiguous().cpu()
    loss_ref, _, _, _, out_ref = ostep.validate_batch(ref, img, heat, c, s, r, g, nm)
    assert rel_rms(output.cpu(), out_ref) < 0.15, rel_rms(output.cpu(), out_ref)
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 0.05


def test_validate_loop_and_prediction_export(tmp_path):
    import scipy.io
    from types import SimpleNamespace
    from pose_adv_aug_amd.stack_hg import validate
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.checkpoint import Checkpoint
    B = 2
    net = create_hg(2, 1, 16, 128, res=256, default_batch=B)
    net.reset_parameters(seed=1)
    batches = [DeviceBatch.synthetic(B, seed=k) for k in range(3)]
    lines = []
    vl, vp, predictions = validate(batches, net, Augmenter(seed=2), 0, SimpleNamespace(print_freq=1), log=lines.append)
    assert predictions.shape == (3 * B, 16, 2) and len(lines) == 3 and np.isfinite(vl) and 0.0 <= vp <= 1.0
    # integer-valued original-image pixels (Evaluation.final_preds: astype(int) + 1)
    assert torch.equal(predictions, predictions.round())
    ck = Checkpoint(); ck.save_prefix = str(tmp_path) + '/'
    ck.save_preds(predictions)                                               # utils/checkpoint.py:38-43
    back = scipy.io.loadmat(str(tmp_path / 'preds.mat'))['preds']
    assert back.shape == (3 * B, 16, 2) and np.array_equal(back, predictions.numpy())
