"""The three stages of the reference as runnable programs with its directory / checkpoint layout (README.md:39-58):
stack-hg.py -> pretrain-s-r-agent.py -> joint-train-pose-s-r-agent.py, each loading what the previous one saved."""
import os
import shutil

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_three_stages_chain_and_the_joint_stage_resumes(tmp_path):
    from pose_adv_aug_amd import stack_hg, pretrain_s_r_agent, joint_train_pose_s_r_agent as J
    from pose_adv_aug_amd.utils.checkpoint import Checkpoint
    exp, eid = str(tmp_path), 'run'
    base = ['--exp_dir', exp, '--exp_id', eid, '--bs', '2', '--print_freq', '1', '--data_dir', str(tmp_path / 'nodata')]
    root = os.path.join(exp, eid)
    # stage 1 (stack-hg.py:40-110): one epoch, checkpoint + predictions
    stack_hg.main(base + ['--is_train', '1', '--nEpochs', '1'])
    assert os.path.isfile(os.path.join(root, 'lr-0.00025-0.pth.tar')) and os.path.isfile(os.path.join(root, 'lr-0.00025-0-preds.mat'))
    assert 'epoch:0, iters:0/4' in open(os.path.join(root, 'train-log.txt')).read()     # (no -model-best copy: val PCKh 0 is not > best_pckh 0, utils/util.py:24)
    # validate-only branch (stack-hg.py:90-96): predictions of the loaded checkpoint
    stack_hg.main(base + ['--load_prefix_pose', 'lr-0.00025-0.pth.tar'])
    assert os.path.isfile(os.path.join(root, 'lr-0.00025-0-preds.mat'))
    # stage 2 (pretrain-s-r-agent.py): distributions + agent checkpoint with an ASNTrainHistory under <sr_dir>-<pose checkpoint>/
    pretrain_s_r_agent.main(base + ['--is_train', '1', '--nEpochs', '1', '--load_prefix_pose', 'lr-0.00025-0.pth.tar'])
    sr = os.path.join(root, 'sr-dir-lr-0.00025-0')
    assert sorted(f for f in os.listdir(sr) if f.endswith('.txt')) == ['train_rotations.txt', 'train_scales.txt', 'training-summary.txt', 'val_rotations.txt', 'val_scales.txt']
    assert open(os.path.join(sr, 'training-summary.txt')).read().splitlines()[0] == 'Epoch\tLR\tTrain Loss\tVal Loss\t'        # pretrain-s-r-agent.py:120-122
    ck = torch.load(os.path.join(sr, 'lr-0.00025-0.pth.tar'), map_location='cpu', weights_only=False)
    assert 'lowest_loss' in ck['train_history'] and ck['train_history']['epoch'][-1]['epoch'] == 0          # what stage 3 loads (joint-...:97-106)
    assert all(k.startswith('module.') for k in ck['state_dict'])
    # stage 3 (joint-train-pose-s-r-agent.py:38-193): starts from both checkpoints at pose epoch 1, one epoch
    args3 = base + ['--is_train', '1', '--load_prefix_pose', 'lr-0.00025-0.pth.tar', '--load_prefix_sr', 'lr-0.00025-0.pth.tar']
    J.main(args3 + ['--nEpochs', '2'])
    jd = os.path.join(root, 'joint-lr-0.00025-0')
    files = sorted(os.listdir(jd))
    assert 'pose-lr-0.00025-1.pth.tar' in files and 'pose-lr-0.00025-1-preds.mat' in files and 'agent-lr-0.00005-1.pth.tar' in files, files
    summary = open(os.path.join(jd, 'pose-training-summary.txt')).read().splitlines()
    assert summary[0] == 'Epoch\tLR\tTrain Loss\tVal Loss\tTrain PCKh\tVal PCKh\t' and summary[1].startswith('1.000000\t0.000250\t')
    log = open(os.path.join(jd, 'train-log.txt')).read()
    assert 'loss_hg_regular' in log and 'loss_hg_sr' in log and 'loss_agent_sr' in log
    pose1 = torch.load(os.path.join(jd, 'pose-lr-0.00025-1.pth.tar'), map_location='cpu', weights_only=False)
    agent1 = torch.load(os.path.join(jd, 'agent-lr-0.00005-1.pth.tar'), map_location='cpu', weights_only=False)
    assert [e['epoch'] for e in pose1['train_history']['epoch']] == [0, 1] and [e['epoch'] for e in agent1['train_history']['epoch']] == [0, 1]
    assert agent1['train_history']['lr'][-1]['lr'] == 5e-5 and 'train_pckh' in pose1['train_history']['pckh'][-1]
    # resume (:72-74, :99-101): the reference derives the directory from the NEW pose prefix, so the files of the interrupted
    # run are expected under <joint_dir>-<that prefix>; continue for one more epoch from there
    jd2 = os.path.join(root, 'joint-pose-lr-0.00025-1')
    shutil.copytree(jd, jd2)
    J.main(base + ['--is_train', '1', '--load_checkpoint', '1', '--load_prefix_pose', 'pose-lr-0.00025-1.pth.tar',
                   '--load_prefix_sr', 'agent-lr-0.00005-1.pth.tar', '--nEpochs', '3'])
    pose2 = torch.load(os.path.join(jd2, 'pose-lr-0.00025-2.pth.tar'), map_location='cpu', weights_only=False)
    agent2 = torch.load(os.path.join(jd2, 'agent-lr-0.00005-2.pth.tar'), map_location='cpu', weights_only=False)
    assert [e['epoch'] for e in pose2['train_history']['epoch']] == [0, 1, 2] and [e['epoch'] for e in agent2['train_history']['epoch']] == [0, 1, 2]
    # the resumed run started from the saved state: optimizer statistics were restored, parameters moved on from the checkpoint
    k = 'module.conv1.weight'
    assert not torch.equal(pose2['state_dict'][k], pose1['state_dict'][k])
    assert pose2['optimizer']['state'][0]['step'] > pose1['optimizer']['state'][0]['step']
    # round trip of a checkpoint through Checkpoint.load_checkpoint into fresh modules reproduces the saved tensors
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.utils.util import PoseTrainHistory
    net = create_hg(2, 1, 16, 256, default_batch=2); opt = RMSprop(net); hist = PoseTrainHistory()
    c = Checkpoint(); c.load_prefix = os.path.join(jd2, 'pose-lr-0.00025-2')
    assert c.load_checkpoint(net, opt, hist)
    sd = net.state_dict(prefix='module.')
    assert all(torch.equal(sd[k].cpu(), v) for k, v in pose2['state_dict'].items() if not k.endswith('num_batches_tracked'))
    assert torch.equal(opt.state_dict()['state'][0]['square_avg'].cpu(), pose2['optimizer']['state'][0]['square_avg']) and hist.epoch[-1]['epoch'] == 2
