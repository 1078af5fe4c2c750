"""RCCL itself: ONE rank through the distributed branch (POSEADV_FORCE_DIST=1, backend nccl = RCCL on ROCm) on the single GPU
of the test box.  librccl is loaded, `dist.all_reduce` runs on the engine's flat gradient buffer and -- with overlap=True -- on
the bucket ranges behind the engine's bucket events on a communication stream.  With one rank a sum is the identity: the result
must equal the plain single-process steps bit for bit.  (Replaces nn.DataParallel, stack-hg.py:49, joint-train-pose-s-r-agent.py:62,90.)"""
import os

import pytest
import torch
import torch.multiprocessing as mp

from tests._rendezvous import file_init_method, set_env

pytestmark = pytest.mark.gpu


def _steps(overlap, dist_on, n=3):
    from pose_adv_aug_amd.stack_hg import broadcast_parameters, train_step
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter, DeviceBatch
    B = 2
    net = create_hg(2, 1, 16, 128, default_batch=B)
    net.reset_parameters(seed=3)
    if dist_on:
        broadcast_parameters(net)
    opt = RMSprop(net, lr=2.5e-4, overlap=overlap)
    aug = Augmenter(seed=11)
    batch = DeviceBatch.synthetic(B, seed=77)
    net.train()
    losses = [float(train_step(net, opt, aug, batch)[0]) for _ in range(n)]
    torch.cuda.synchronize()
    return net.flat_params.cpu(), net.flat_grads.cpu(), losses


def _worker(rank, init, out):
    set_env(0, 1, init, local_rank=0, POSEADV_FORCE_DIST='1')
    os.environ.pop('POSEADV_DIST_BACKEND', None)                       # default on a GPU box: nccl
    import torch.distributed as dist
    from pose_adv_aug_amd.stack_hg import init_distributed
    init_distributed()
    assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    calls = []
    orig = dist.all_reduce

    def counted(t, *a, **k):
        calls.append((t.numel(), bool(k.get('async_op', False)), t.is_cuda))
        return orig(t, *a, **k)
    dist.all_reduce = counted
    res = {}
    for overlap in (False, True):
        calls.clear()
        p, g, losses = _steps(overlap, True)
        res[overlap] = dict(params=p, grads=g, losses=losses, calls=list(calls))
    maps = open('/proc/self/maps').read()
    res['rccl_loaded'] = 'librccl' in maps
    out['r'] = res
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_through_rccl_equals_the_plain_steps():
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    mp.spawn(_worker, args=(file_init_method(), out), nprocs=1, join=True)
    r = out['r']
    assert r['rccl_loaded']
    plain_p, plain_g, plain_l = _steps(False, False)
    for overlap in (False, True):
        assert torch.equal(r[overlap]['params'], plain_p) and torch.equal(r[overlap]['grads'], plain_g)
        # (the reported loss is a sum of per-workgroup float atomics: same value up to the order of the additions)
        assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(r[overlap]['losses'], plain_l))
        assert all(c[2] for c in r[overlap]['calls'])                                  # device buffers went to the collective
    n = plain_p.numel()
    whole = [c for c in r[False]['calls'] if c[0] == n]
    assert len(whole) == 3 and not any(c[1] for c in whole)                            # ONE all-reduce of the flat gradient per step
    per_step = [c for c in r[True]['calls'] if c[0] < n]
    assert len([c for c in per_step if c[1]]) == 3 * 2                                 # overlapped: one asynchronous bucket per stack and step
    assert sum(c[0] for c in per_step) == 3 * n                                        # every gradient exchanged exactly once per step
