"""Data parallelism with the REAL engine: two processes on the one GPU of the test box (gloo carries the all-reduce; on a
multi-GPU node the same code runs one process per GPU over RCCL).  SURVEY.md section 8e: per-rank BatchNorm statistics,
ONE all-reduce of the flat gradient per step, 1/world folded into the fused RMSprop, bit-identical replicas."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from tests._rendezvous import engine_rank_env, file_init_method, set_env

pytestmark = pytest.mark.gpu

if os.environ.get('POSEADV_TEST_DIST_BACKEND') == 'nccl' and torch.cuda.device_count() < 2:
    pytestmark = [pytest.mark.gpu, pytest.mark.skip(reason='two ranks over RCCL need two GPUs (a GPU holds one RCCL rank)')]


def _worker(rank, world, init, out, overlap=False, stacks=1):
    set_env(rank, world, init, **engine_rank_env(rank))
    import torch.distributed as dist
    from pose_adv_aug_amd.stack_hg import init_distributed, broadcast_parameters, train_step
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    init_distributed()
    B = 2
    net = create_hg(stacks, 1, 16, 128, default_batch=B)
    net.reset_parameters(seed=100 + rank)                      # replicas start DIFFERENT: the broadcast must fix that
    broadcast_parameters(net)
    start = net.flat_params.clone()
    opt = RMSprop(net, lr=2.5e-4, overlap=overlap)
    aug = Augmenter(seed=50 + rank)
    batch = DeviceBatch.synthetic(B, seed=900 + rank)
    net.train()
    losses = []
    calls = []
    if overlap:
        orig = dist.all_reduce

        def counted(t, *a, **k):
            calls.append((t.numel(), bool(k.get('async_op', False))))
            return orig(t, *a, **k)
        dist.all_reduce = counted
    for i in range(2):
        loss, _, _ = train_step(net, opt, aug, batch)
        losses.append(float(loss))
    torch.cuda.synchronize()
    out[rank] = dict(start=start.cpu(), params=net.flat_params.cpu(), grads=net.flat_grads.cpu(), buffers=net.flat_buffers.cpu(), losses=losses,
                     calls=calls, nparams=net.flat_params.numel())
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_data_parallel_step_on_the_engine():
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, file_init_method(), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert torch.equal(r0['start'], r1['start'])                                  # broadcast: rank 0's initialisation everywhere
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['grads'], r1['grads'])      # identical replicas, summed gradient
    assert not torch.equal(r0['buffers'], r1['buffers'])                          # BatchNorm running statistics are per rank (DataParallel semantics)
    assert r0['losses'] != r1['losses']                                           # different shards, different augmentations
    # the same two steps in ONE process: both shards' gradients from the same weights, averaged, one RMSprop step each
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    nets = []
    for rank in range(2):
        n = create_hg(1, 1, 16, 128, default_batch=2); n.reset_parameters(seed=100); n.train()
        nets.append((n, Augmenter(seed=50 + rank), DeviceBatch.synthetic(2, seed=900 + rank)))
    master = nets[0][0]
    opt = RMSprop(master, lr=2.5e-4)
    assert torch.equal(master.flat_params.cpu(), r0['start'])
    for i in range(2):
        gs = []
        for n, aug, batch in nets:
            if n is not master:
                n.flat_params.copy_(master.flat_params); n.weights_changed()
            data = aug.regular(batch)
            n.loss_and_backward(img4=data['img4'], pts=data['pts'])
            gs.append(n.flat_grads.clone())
        master.flat_grads.copy_((gs[0] + gs[1]) * 0.5)
        opt.step()
    torch.cuda.synchronize()
    assert torch.equal(master.flat_params.cpu(), r0['params'])                    # bit for bit the data-parallel result


def test_overlapped_gradient_exchange_gives_the_same_result():
    """RMSprop(overlap=True): the backward pass runs in phases (pa_hg_backward_phase); each stack's hourglass gradients are
    all-reduced on a communication stream behind the engine's bucket event (pa_hg_bucket_wait: main chain + weight-gradient
    stream) while the earlier stacks still run, the remaining ranges at the end.  Same parameters, gradients and statistics,
    bit for bit, as the single all-reduce after the backward pass; per step 2 asynchronous buckets + 3 remaining ranges."""
    ctx = mp.get_context('spawn')
    res = {}
    for overlap in (False, True):
        out = ctx.Manager().dict()
        mp.spawn(_worker, args=(2, file_init_method(), out, overlap, 2), nprocs=2, join=True)
        res[overlap] = (out[0], out[1])
    a0, a1 = res[False]
    b0, b1 = res[True]
    assert torch.equal(b0['params'], b1['params'])
    assert torch.equal(a0['params'], b0['params']) and torch.equal(a0['grads'], b0['grads']) and torch.equal(a0['buffers'], b0['buffers'])
    per_step = len(b0['calls']) // 2
    step = b0['calls'][:per_step]
    assert [c for c in step if c[1]] and len([c for c in step if c[1]]) == 2          # one asynchronous bucket per stack
    assert sum(c[0] for c in step) == b0['nparams']                                    # every gradient exchanged exactly once


def _joint_worker(rank, world, init, exp_dir, out):
    """joint_train_pose_s_r_agent.main() on `world` ranks with the REAL engine (both on the one GPU, gloo collectives)."""
    set_env(rank, world, init, **engine_rank_env(rank))
    import torch.distributed as dist
    from pose_adv_aug_amd import joint_train_pose_s_r_agent as J
    from pose_adv_aug_amd.models import asn_stacked_hg as M
    reduces, bins, saves, nets = [], [], [], {}
    orig_ar, orig_bins, orig_save = dist.all_reduce, J.sample_bins, torch.save

    def all_reduce(t, *a, **k):
        reduces.append(t.numel())
        return orig_ar(t, *a, **k)

    def sample_bins(logits, seed, step, slot):
        probs, idx = orig_bins(logits, seed, step, slot)
        bins.append((int(step), int(slot), idx.cpu().tolist()))
        return probs, idx

    def save(obj, path, *a, **k):
        saves.append(os.path.basename(path))
        return orig_save(obj, path, *a, **k)
    dist.all_reduce, J.sample_bins, torch.save = all_reduce, sample_bins, save
    for name in ('create_hg', 'create_asn'):
        def wrap(f=getattr(M, name), name=name):
            def g(*a, **k):
                nets[name] = f(*a, **k)
                return nets[name]
            return g
        setattr(M, name, wrap())
    J.main(['--exp_dir', exp_dir, '--exp_id', 'run', '--bs', '2', '--print_freq', '1', '--data_dir', os.path.join(exp_dir, 'nodata'),
            '--is_train', '1', '--load_prefix_pose', 'lr-0.00025-0.pth.tar', '--load_prefix_sr', 'lr-0.00025-0.pth.tar', '--nEpochs', '2'])
    torch.cuda.synchronize()
    hg, asn = nets['create_hg'], nets['create_asn']
    out[rank] = dict(reduces=reduces, bins=bins, saves=saves, hg=hg.flat_params.cpu(), asn=asn.flat_params.cpu(),
                     hg_buf=hg.flat_buffers.cpu(), n_hg=hg.flat_params.numel(), n_asn=asn.flat_params.numel())
    dist.barrier()
    dist.destroy_process_group()


def test_joint_stage_main_on_two_ranks(tmp_path):
    """BASELINE configs[3] is a data-parallel configuration (joint-train-pose-s-r-agent.py:62,90 wrap BOTH nets in DataParallel):
    main() of the joint stage on 2 ranks.  Both replicas of both nets are broadcast from rank 0 and stay bit-identical; per pose
    step ONE all-reduce of the pose net's flat gradient, per epoch ONE of the agent's (train_agent_sr: one batch); the ranks draw
    their own bins from their own shards (seed, step counters shared, logits differ); rank 0 alone writes pose-* / agent-* files."""
    from pose_adv_aug_amd import stack_hg, pretrain_s_r_agent
    exp = str(tmp_path)
    base = ['--exp_dir', exp, '--exp_id', 'run', '--bs', '2', '--print_freq', '1', '--data_dir', str(tmp_path / 'nodata')]
    stack_hg.main(base + ['--is_train', '1', '--nEpochs', '1'])                         # the two checkpoints stage 3 starts from
    pretrain_s_r_agent.main(base + ['--is_train', '1', '--nEpochs', '1', '--load_prefix_pose', 'lr-0.00025-0.pth.tar'])
    torch.cuda.synchronize()
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    mp.spawn(_joint_worker, args=(2, file_init_method(), exp, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert torch.equal(r0['hg'], r1['hg']) and torch.equal(r0['asn'], r1['asn'])       # identical replicas of both nets at the end
    assert not torch.equal(r0['hg_buf'], r1['hg_buf'])                                  # per-rank BatchNorm statistics
    for r in (r0, r1):
        big = [n for n in r['reduces'] if n in (r['n_hg'], r['n_asn'])]
        # every rank has its own 4 synthetic batches: 4 pose steps, then the epoch's single agent update
        assert big == [r['n_hg']] * 4 + [r['n_asn']], big
        assert all(n <= 64 or n in (r['n_hg'], r['n_asn']) for n in r['reduces'])       # everything else: the meters' tables
    steps0 = [(s, sl) for s, sl, _ in r0['bins']]
    assert steps0 == [(s, sl) for s, sl, _ in r1['bins']] and len(steps0) == 6          # the two odd pose steps (2 draws each) + the agent update (2 draws)
    assert [b for _, _, b in r0['bins']] != [b for _, _, b in r1['bins']]               # own shards -> own logits -> own bins
    assert sorted(f for f in r0['saves'] if f.endswith('.pth.tar')) == ['agent-lr-0.00005-1.pth.tar', 'pose-lr-0.00025-1.pth.tar'] and r1['saves'] == []
    jd = os.path.join(exp, 'run', 'joint-lr-0.00025-0')
    assert os.path.isfile(os.path.join(jd, 'pose-lr-0.00025-1-preds.mat')) and 'loss_agent_sr' in open(os.path.join(jd, 'train-log.txt')).read()


@pytest.mark.parametrize('overlap', [0, 1])
def test_bench_line_from_two_ranks(overlap):
    """bench.py's own N > 1 path as the driver launches it (torch.distributed.run, one JSON line from rank 0), two ranks on the one
    GPU of the test box with gloo carrying the exchange -- including the overlapped exchange, whose collectives must not be entered
    by rank 0 alone in the untimed legs (roofline pass, PCKh parity)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, POSEADV_DIST_BACKEND=os.environ.get('POSEADV_TEST_DIST_BACKEND', 'gloo'))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'POSEADV_DIST_INIT', 'POSEADV_FORCE_DIST'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--overlap', str(overlap)], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 alone prints
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['config']['global_batch'] == 48 and d['config']['parallelism'] == ('dp2+overlapped-exchange' if overlap else 'dp2')
    assert d['pckh_parity']['match'] is True and d['roofline']['frac'] > 0 and d['cpu_baseline'] is None


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as a PLAIN command (no torch.distributed.run, no WORLD_SIZE in the environment): bench.py starts its
    two ranks itself over a file-store rendezvous (bench.self_launch) -- what the driver's multi-GPU scaling run invokes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEADV_DIST_BACKEND=os.environ.get('POSEADV_TEST_DIST_BACKEND', 'gloo'))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'POSEADV_DIST_INIT', 'POSEADV_FORCE_DIST'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                        '--no-traffic'], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 48 and d['config']['parallelism'] == 'dp2' and d['value'] > 0
