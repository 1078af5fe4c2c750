"""Data parallelism with the REAL engine: two processes on the one GPU of the test box (gloo carries the all-reduce; on a
multi-GPU node the same code runs one process per GPU over RCCL).  SURVEY.md section 8e: per-rank BatchNorm statistics,
ONE all-reduce of the flat gradient per step, 1/world folded into the fused RMSprop, bit-identical replicas."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out, overlap=False, stacks=1):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      POSEADV_DIST_BACKEND='gloo')
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
    port = 29500 + ((os.getpid() + 13) % 500)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
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
        port = 29500 + ((os.getpid() + 29 + int(overlap)) % 500)
        mp.spawn(_worker, args=(2, port, out, overlap, 2), nprocs=2, join=True)
        res[overlap] = (out[0], out[1])
    a0, a1 = res[False]
    b0, b1 = res[True]
    assert torch.equal(b0['params'], b1['params'])
    assert torch.equal(a0['params'], b0['params']) and torch.equal(a0['grads'], b0['grads']) and torch.equal(a0['buffers'], b0['buffers'])
    per_step = len(b0['calls']) // 2
    step = b0['calls'][:per_step]
    assert [c for c in step if c[1]] and len([c for c in step if c[1]]) == 2          # one asynchronous bucket per stack
    assert sum(c[0] for c in step) == b0['nparams']                                    # every gradient exchanged exactly once
