"""world_size-2 (gloo, CPU) test of the data-parallel exchange: ONE all-reduce (sum) of the flat gradient,
1/world scaling folded into the optimizer, replicas stay bit-identical, parameters are broadcast once."""
import os
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import step as ostep


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.stack_hg import broadcast_parameters
    n = 1000
    g = torch.Generator().manual_seed(7)
    net = types.SimpleNamespace(flat_params=torch.randn(n, generator=g) + rank,        # replicas start different
                                flat_grads=torch.zeros(n), flat_buffers=torch.zeros(4) + rank,
                                _table=[('w', (n,), 0, n, 0)], _ensure_table=lambda: None,
                                weights_changed=lambda: None)
    broadcast_parameters(net)                                                           # rank 0's values everywhere
    opt = RMSprop(net)
    local = torch.Generator().manual_seed(100 + rank)
    per_rank = torch.randn(n, generator=local)                                          # this rank's shard gradient
    net.flat_grads.copy_(per_rank)
    gscale = opt.allreduce_grads()
    # emulate the fused device update with the oracle's RMSprop on (sum * gscale)
    v = torch.zeros(n)
    ostep.rmsprop_update(net.flat_params, net.flat_grads * gscale, v, 2.5e-4)
    out[rank] = (net.flat_params.clone(), net.flat_grads.clone(), gscale, net.flat_buffers.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    p0, g0, s0, b0 = out[0]
    p1, g1, s1, b1 = out[1]
    assert s0 == s1 == 0.5
    assert torch.equal(g0, g1) and torch.equal(p0, p1) and torch.equal(b0, b1)         # identical replicas
    ref = sum(torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(2))
    assert torch.allclose(g0, ref)
    # equals the single-process step on the mean gradient of the concatenated batch
    pref = torch.randn(1000, generator=torch.Generator().manual_seed(7))
    ostep.rmsprop_update(pref, ref * 0.5, torch.zeros(1000), 2.5e-4)
    assert torch.allclose(p0, pref)
