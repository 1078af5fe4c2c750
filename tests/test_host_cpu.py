"""CPU-only checks of the host layer: the C-ABI library loads and exports every symbol that
include/poseadv.h declares (no compute calls without a GPU), option parsing, checkpoint file naming /
round trip, histories, LR schedule and the vectorised reward shaping against the oracle."""
import os
import re
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import pose_adv_aug_amd as P
    if not os.path.isfile(P.LIB_PATH):
        P.build()
    hdr = open(os.path.join(ROOT, 'include', 'poseadv.h')).read()
    declared = sorted(set(re.findall(r'\b(pa_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 35
    l = P.lib()
    missing = [n for n in declared if not hasattr(l, n)]
    assert not missing, missing
    assert l.pa_version() >= 1
    # every symbol the Python layer binds is declared in the header
    assert set(P.EXPORTS) <= set(declared), sorted(set(P.EXPORTS) - set(declared))


def test_no_cpu_fallback_without_gpu():
    import pose_adv_aug_amd as P
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    net = create_hg(2, 1, 16, 256)
    with pytest.raises(P.PoseAdvError):
        net(torch.zeros(1, 3, 256, 256))
    with pytest.raises(P.PoseAdvError):
        from pose_adv_aug_amd.pylib import HumanPts
        HumanPts.pts2heatmap(np.ones((16, 2)), [64, 64])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'pose_adv_aug_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py') or f.endswith('.hip') or f.endswith('.h'):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(d, f)
                assert 'from tests' not in src, os.path.join(d, f)


def test_options_and_prefix_normalisation(tmp_path, capsys):
    from pose_adv_aug_amd.options.train_options import TrainOptions
    opt = TrainOptions().parse(['--exp_dir', str(tmp_path), '--exp_id', 'e1', '--bs', '24', '--is_train', 'false',
                                '--load_prefix_pose', 'lr-0.00025-10.pth.tar'])
    assert opt.bs == 24 and opt.lr == 2.5e-4 and opt.agent_lr == 5e-5 and opt.nEpochs == 100 and opt.print_freq == 10
    assert opt.is_train is True                       # type=bool quirk of the reference: any non-empty string is True
    assert opt.load_prefix_pose == 'lr-0.00025-10-'   # what options/base_options.py:62-65 intends
    assert opt.load_prefix_pose[0:-1] == 'lr-0.00025-10'          # stack-hg.py:59 strips the '-' again
    assert os.path.isfile(os.path.join(str(tmp_path), 'e1', 'opt.txt'))
    with pytest.raises(SystemExit):
        TrainOptions().parse(['--exp_dir', str(tmp_path)])       # missing --exp_id


def test_histories_meter_and_lr_schedule():
    from pose_adv_aug_amd.utils.util import PoseTrainHistory, ASNTrainHistory, AverageMeter, adjust_lr
    from oracle import pylib as opl
    h = PoseTrainHistory()
    h.update({'epoch': 0}, {'lr': 2.5e-4}, {'train_loss': 1., 'val_loss': 1.}, {'train_pckh': .1, 'val_pckh': .2})
    assert h.is_best and h.best_pckh == .2
    h.update({'epoch': 1}, {'lr': 2.5e-4}, {'train_loss': 1., 'val_loss': 1.}, {'train_pckh': .1, 'val_pckh': .1})
    assert not h.is_best
    h2 = PoseTrainHistory(); h2.load_state_dict(h.state_dict())
    assert h2.epoch[-1]['epoch'] == 1 and h2.best_pckh == .2
    a = ASNTrainHistory(); a.update({'epoch': 0}, {'lr': 5e-5}, {'train_loss': 3.0})
    assert a.is_best and a.lowest_loss == 3.0
    m = AverageMeter(); m.update(2.0); m.update(4.0)
    assert m.avg == 3.0 and m.val == 4.0
    opt = types.SimpleNamespace(lr=2.5e-4)
    o = types.SimpleNamespace(param_groups=[{'lr': 2.5e-4}])
    lr = 2.5e-4
    for epoch in range(0, 150):
        adjust_lr(opt, o, epoch)
        lr = opl.adjust_lr_value(lr, epoch)
        assert abs(o.param_groups[0]['lr'] - lr) < 1e-12
    assert abs(lr - 2.5e-4 * 0.2 * 0.5) < 1e-12


def test_gen_groundtruth_matches_oracle():
    from pose_adv_aug_amd.utils.util import gen_groundtruth
    from oracle import pylib as opl
    from tests.test_oracle_golden import load, t
    g = load('pylib.npz')
    out = gen_groundtruth(t(g['gg_p']), t(g['gg_idx']), t(g['gg_reg']), t(g['gg_agent']))
    assert np.allclose(out.numpy(), g['gg_out'], atol=1e-7)
    rng = np.random.default_rng(0)
    for _ in range(100):
        p = torch.softmax(torch.tensor(rng.normal(0, 2.5, (24, 7)), dtype=torch.float32), 1)
        idx = torch.tensor(rng.integers(0, 7, (24, 1)))
        a = torch.tensor(rng.random(24), dtype=torch.float32); b = torch.tensor(rng.random(24), dtype=torch.float32)
        assert torch.allclose(gen_groundtruth(p, idx, a, b), opl.gen_groundtruth(p, idx, a, b), atol=1e-6)


class _FakeNet(object):
    """state_dict / load_state_dict surface of the engine's modules on CPU tensors (checkpoint plumbing only)."""

    def __init__(self):
        self.w = {'conv1.weight': torch.arange(12.).view(2, 2, 1, 3), 'bn1.running_mean': torch.zeros(2)}

    def state_dict(self, prefix=''):
        return {prefix + k: v for k, v in self.w.items()}

    def load_state_dict(self, sd, strict=True):
        unexpected = []
        for k, v in sd.items():
            kk = k[7:] if k.startswith('module.') else k
            if kk in self.w:
                self.w[kk] = v.clone()
            else:
                unexpected.append(k)
        return [], unexpected


def test_checkpoint_names_and_round_trip(tmp_path):
    from pose_adv_aug_amd.utils.checkpoint import Checkpoint
    from pose_adv_aug_amd.utils.util import PoseTrainHistory
    net = _FakeNet()
    opt = types.SimpleNamespace(state_dict=lambda: {'state': {}, 'param_groups': [{'lr': 2.5e-4}]}, load_state_dict=lambda sd: None)
    h = PoseTrainHistory()
    h.update({'epoch': 10}, {'lr': 2.5e-4}, {'train_loss': 1., 'val_loss': 1.}, {'train_pckh': .1, 'val_pckh': .2})
    ck = Checkpoint(); ck.save_prefix = str(tmp_path) + '/'
    path = ck.save_checkpoint(net, opt, h, torch.zeros(3, 16, 2))
    assert os.path.basename(path) == 'lr-0.00025-10.pth.tar'                      # utils/checkpoint.py:15-16
    for f in ('lr-0.00025-10-preds.mat', 'lr-0.00025-10-model-best.pth.tar', 'lr-0.00025-10-preds-best.mat'):
        assert os.path.isfile(os.path.join(str(tmp_path), f)), f
    saved = torch.load(path, weights_only=False)
    assert list(saved['state_dict'].keys())[0] == 'module.conv1.weight'            # DataParallel prefix kept
    saved['state_dict']['module.not_in_net'] = torch.zeros(1)
    torch.save(saved, path)
    net2 = _FakeNet(); net2.w['conv1.weight'] = torch.zeros(2, 2, 1, 3)
    h2 = PoseTrainHistory()
    ck2 = Checkpoint(); ck2.load_prefix = os.path.join(str(tmp_path), 'lr-0.00025-10')
    assert ck2.load_checkpoint(net2, opt, h2)
    assert torch.equal(net2.w['conv1.weight'], net.w['conv1.weight']) and h2.epoch[-1]['epoch'] == 10
    assert not Checkpoint().load_checkpoint(net2)                                   # missing file: message, no exception


def test_text_log_formats(tmp_path, capsys):
    """utils/visualizer.py:66-87 (print_log / write_log) and utils/logger.py:24-74 (Logger) byte for byte."""
    from collections import OrderedDict
    from pose_adv_aug_amd.utils.visualizer import Visualizer
    from pose_adv_aug_amd.utils.logger import Logger
    v = Visualizer(log_path=str(tmp_path / 'exp' / 'log.txt'))
    d = OrderedDict([('loss', 0.01234567), ('pckh', 0.5), ('pckh_origin_res', 1.0)])
    m0 = v.print_log(3, 0, 2, value1=d)
    m1 = v.print_log(3, 1, 2, value1=d, value2=OrderedDict([('t', 1.23456)]))
    assert m0 == 'epoch:3, iters:0/2 loss: 0.0123 pckh: 0.5000 pckh_origin_res: 1.0000 '
    assert m1 == m0.replace('iters:0/2', 'iters:1/2') + '\nt:1.235 \n##########################################'
    assert capsys.readouterr().out == m0 + '\n' + m1 + '\n'
    assert (tmp_path / 'exp' / 'log.txt').read_text() == m0 + '\n' + m1 + '\n'
    path = str(tmp_path / 'training-summary.txt')
    lg = Logger(path, title='training-summary')
    lg.set_names(['Epoch', 'LR', 'Train Loss', 'Val Loss'])
    lg.append([0, 2.5e-4, 0.5, 0.25]); lg.append([1, 2.5e-4, 0.125, 0.1]); lg.close()
    assert open(path).read() == 'Epoch\tLR\tTrain Loss\tVal Loss\t\n0.000000\t0.000250\t0.500000\t0.250000\t\n1.000000\t0.000250\t0.125000\t0.100000\t\n'
    lg2 = Logger(path, resume=True)
    assert lg2.names == ['Epoch', 'LR', 'Train Loss', 'Val Loss'] and lg2.numbers['Train Loss'] == ['0.500000', '0.125000']
    lg2.close()


def test_device_meters_see_every_iteration_split_by_kind():
    """utils.util.DeviceMeters (CPU tensors here): the six meters of train_hg (joint-train-pose-s-r-agent.py:197-305) get
    every step's value, regular and agent-augmented steps separately -- not a print_freq subsample."""
    from pose_adv_aug_amd.utils.util import DeviceMeters
    from pose_adv_aug_amd.joint_train_pose_s_r_agent import METER_NAMES
    m = DeviceMeters(METER_NAMES, torch.device('cpu'))
    for i in range(7):
        loss, pckh = torch.tensor(float(i)), torch.tensor(i / 10.0)
        tag = ('loss_hg_regular', 'pckhs_regular') if i % 2 == 0 else ('loss_hg_sr', 'pckhs_sr')
        m.update({'loss_hg': loss, 'pckh': pckh, tag[0]: loss, tag[1]: pckh})
    d = m.averages()
    assert list(d) == list(METER_NAMES)
    assert d['loss_hg'] == 3.0 and d['loss_hg_regular'] == 3.0 and d['loss_hg_sr'] == 3.0           # (0+2+4+6)/4, (1+3+5)/3
    assert abs(d["pckhs_sr"] - 0.3) < 1e-7 and abs(d["pckh"] - 0.3) < 1e-7
    assert DeviceMeters(('a',), torch.device('cpu')).averages() == {'a': 0.0}


def test_batch_feed_is_sized_and_reiterable(tmp_path):
    """what len(train_loader) / enumerate(train_loader) need (stack-hg.py:133,183): MPII.batches() is a sized feed"""
    from pose_adv_aug_amd.data import BatchFeed, num_samples
    calls = []

    def gen():
        calls.append(1)
        for k in range(3):
            yield types.SimpleNamespace(B=2, k=k)
    f = BatchFeed(3, 6, gen)
    assert len(f) == 3 and num_samples(f) == 6 and not calls           # asking for sizes consumes nothing
    assert [b.k for b in f] == [0, 1, 2] and [b.k for b in f] == [0, 1, 2] and len(calls) == 2
    assert num_samples([types.SimpleNamespace(B=4), types.SimpleNamespace(B=3)]) == 7


def test_mpii_feed_sizes_and_rank_shards(tmp_path):
    import json
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    anno = [dict(dataset='MPII', isValidation=float(i % 5 == 0), img_paths='i%d.jpg' % i, joint_self=[[1.0, 2.0, 1.0]] * 16,
                 objpos=[100.0, 80.0], scale_provided=1.0, normalizer=10.0) for i in range(53)]
    p = tmp_path / 'a.json'
    p.write_text(json.dumps(anno))
    ds = MPII(str(p), str(tmp_path), is_train=True, log=lambda m: None)
    n = len(ds)
    assert n == 42
    f = ds.batches(8, drop_last=True)
    assert len(f) == 5 and f.num_samples == 40
    f = ds.batches(8, drop_last=False, shuffle=False)
    assert len(f) == 6 and f.num_samples == 42
    f0, f1 = ds.batches(8, drop_last=True, rank=0, world=2), ds.batches(8, drop_last=True, rank=1, world=2)
    assert len(f0) == len(f1) == 2 and f0.num_samples == f1.num_samples == 16       # equal counts on every rank


def test_distribution_files_are_sized_by_the_split_not_by_the_drop_last_feed(tmp_path):
    """stage 2 (pretrain-s-r-agent.py:262-275): a distribution file has one row per person of the SPLIT; the shuffled training
    feed drops its last partial batch (n % bs != 0 on real MPII), so a complete file must not be judged 'cut short'."""
    import json
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    from pose_adv_aug_amd.data import dataset_size, num_samples, BatchFeed
    from pose_adv_aug_amd.pretrain_s_r_agent import distribution_rows
    anno = [dict(dataset='MPII', isValidation=float(i % 5 == 0), img_paths='i%d.jpg' % i, joint_self=[[1.0, 2.0, 1.0]] * 16,
                 objpos=[100.0, 80.0], scale_provided=1.0, normalizer=10.0) for i in range(53)]
    p = tmp_path / 'a.json'
    p.write_text(json.dumps(anno))
    ds = MPII(str(p), str(tmp_path), is_train=True, log=lambda m: None)
    shuffled, ordered = ds.batches(8, drop_last=True), ds.batches(8, shuffle=False, drop_last=False)
    assert num_samples(shuffled) == 40 and dataset_size(shuffled) == 42 == num_samples(ordered) == dataset_size(ordered)
    assert dataset_size(BatchFeed.of([types.SimpleNamespace(B=3), types.SimpleNamespace(B=2)])) == 5
    path = str(tmp_path / 'train_scales.txt')
    calls = []

    def write_rows(n):
        def collect(tmp):
            calls.append(n)
            with open(tmp, 'w') as fd:
                for _ in range(n):
                    fd.write('0.25 0.25 0.50\n')
        return collect
    rows = distribution_rows(path, dataset_size(shuffled), write_rows(42))               # missing: collected once
    assert len(rows) == 42 and calls == [42]
    rows = distribution_rows(path, dataset_size(shuffled), write_rows(42))               # complete: NOT collected again
    assert len(rows) == 42 and calls == [42]
    with open(path, 'w') as fd:                                                          # cut short by an interrupted run
        fd.write('0.25 0.25 0.50\n' * 17)
    rows = distribution_rows(path, dataset_size(shuffled), write_rows(42))
    assert len(rows) == 42 and calls == [42, 42] and not os.path.exists(path + '.collecting')
    with pytest.raises(RuntimeError):                                                    # a collection that comes out short is an error
        distribution_rows(str(tmp_path / 'val_scales.txt'), 11, write_rows(10))


def test_optimizer_state_loads_torch_files_and_warns(tmp_path):
    """utils/optim.RMSprop.load_state_dict: torch.optim.RMSprop files keyed by position OR by arbitrary ids (torch 0.3), a
    missing / mis-shaped entry is reported"""
    import warnings
    from pose_adv_aug_amd.utils.optim import RMSprop
    table = [('a.weight', (2, 3), 0, 6, 0), ('a.bias', (3,), 8, 3, 0), ('bn.running_mean', (3,), 0, 3, 1)]
    net = types.SimpleNamespace(flat_params=torch.zeros(12), flat_grads=torch.zeros(12), _table=table, _ensure_table=lambda: None)
    ps = [torch.nn.Parameter(torch.randn(2, 3)), torch.nn.Parameter(torch.randn(3))]
    topt = torch.optim.RMSprop(ps, lr=1e-3, alpha=0.99, eps=1e-8)
    for q in ps:
        q.grad = torch.randn_like(q)
    topt.step()
    sd = topt.state_dict()
    o = RMSprop(net); o.load_state_dict(sd)
    assert torch.equal(o.square_avg[0:6].view(2, 3), sd['state'][0]['square_avg']) and torch.equal(o.square_avg[8:11], sd['state'][1]['square_avg'])
    assert o.param_groups[0]['lr'] == 1e-3
    sd_ids = {'state': {140001: sd['state'][0], 140777: sd['state'][1]}, 'param_groups': [dict(sd['param_groups'][0], params=[140001, 140777])]}
    o2 = RMSprop(net); o2.load_state_dict(sd_ids)
    assert torch.equal(o2.square_avg, o.square_avg)
    bad = {'state': {0: sd['state'][0]}, 'param_groups': sd['param_groups']}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        RMSprop(net).load_state_dict(bad)
    assert any('no square_avg' in str(x.message) for x in w)
    own = RMSprop(net); own.load_state_dict(o.state_dict())
    assert torch.equal(own.square_avg[0:6], o.square_avg[0:6]) and torch.equal(own.square_avg[8:11], o.square_avg[8:11])


def test_summary_logger_format_and_resume(tmp_path):
    """the pose-training-summary.txt table (joint-train-pose-s-r-agent.py:145-147,175): tab-terminated names, %.6f rows"""
    from pose_adv_aug_amd.utils.logger import Logger
    p = str(tmp_path / 's.txt')
    lg = Logger(p, title='t'); lg.set_names(['Epoch', 'LR']); lg.append([0, 2.5e-4]); lg.close()
    assert open(p).read() == 'Epoch\tLR\t\n0.000000\t0.000250\t\n'
    lg = Logger(p, resume=True)
    assert lg.names == ['Epoch', 'LR'] and lg.numbers['LR'] == ['0.000250']
    lg.append([1, 5e-5]); lg.close()
    assert open(p).read().count('\n') == 3
    with pytest.raises(ValueError):
        l2 = Logger(str(tmp_path / 'x.txt')); l2.set_names(['a']); l2.append([1, 2])


def test_bench_self_launch_propagates_a_failing_rank():
    """bench.py --gpus 2 without a launcher starts its own ranks; on this GPU-less box both ranks fail loudly (no CPU fallback) and the
    launcher returns non-zero instead of hanging or printing a line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if __import__('torch').cuda.is_available():
        pytest.skip('GPU present: covered by tests/test_gpu_dist.py::test_bench_launches_its_own_ranks')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'POSEADV_DIST_INIT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], cwd=root, env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
