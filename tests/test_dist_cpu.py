"""world_size-2 (gloo, CPU) test of the data-parallel exchange: ONE all-reduce (sum) of the flat gradient,
1/world scaling folded into the optimizer, replicas stay bit-identical, parameters are broadcast once."""
import os
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import step as ostep
from tests._rendezvous import file_init_method, set_env


def _worker(rank, world, init, out):
    dist.init_process_group('gloo', init_method=init, rank=rank, world_size=world)
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
    mp.spawn(_worker, args=(2, file_init_method(), out), nprocs=2, join=True)
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


def test_flat_gradient_allreduce_world4():
    """the same exchange typed for N not in {1, 2} (joint-train-pose-s-r-agent.py:62,90 run on 4 / 8 GPUs): 1/4 scaling, replicas
    identical on every rank, equal to one process stepping on the mean of the four shard gradients"""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(4, file_init_method(), out), nprocs=4, join=True)
    assert all(out[r][2] == 0.25 for r in range(4))
    for r in range(1, 4):
        assert torch.equal(out[0][0], out[r][0]) and torch.equal(out[0][1], out[r][1]) and torch.equal(out[0][3], out[r][3])
    ref = sum(torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(4))
    assert torch.allclose(out[0][1], ref, atol=1e-6)
    assert torch.equal(out[0][3], torch.zeros(4))                                       # rank 0's buffers everywhere
    pref = torch.randn(1000, generator=torch.Generator().manual_seed(7))
    ostep.rmsprop_update(pref, ref * 0.25, torch.zeros(1000), 2.5e-4)
    assert torch.allclose(out[0][0], pref)


def _main_worker(rank, world, init, exp_dir, out):
    """stack_hg.main() -- the REAL host control flow -- on 2 gloo ranks with the engine stubbed at the C ABI."""
    set_env(rank, world, init)
    from tests import abi_stub
    fake = abi_stub.install(rank)
    orig_ar, orig_save = dist.all_reduce, torch.save
    saves = []

    def all_reduce(t, op=dist.ReduceOp.SUM, **kw):
        fake.log.append(('all_reduce', (t.numel(), str(t.dtype))))
        return orig_ar(t, op=op, **kw)

    def save(obj, path, *a, **kw):
        saves.append(os.path.basename(path))
        return orig_save(obj, path, *a, **kw)
    dist.all_reduce, torch.save = all_reduce, save
    from pose_adv_aug_amd import stack_hg
    captured = {}
    orig_create = stack_hg.create_hg

    def create(*a, **kw):
        captured['net'] = orig_create(*a, **kw)
        return captured['net']
    stack_hg.create_hg = create
    stack_hg.main(['--exp_dir', exp_dir, '--exp_id', 'dp', '--bs', '2', '--nEpochs', '1', '--is_train', '1', '--print_freq', '2'])
    net = captured['net']
    out[rank] = dict(log=[(n, a if n in ('pa_sample_aug', 'pa_rmsprop_step', 'all_reduce', 'pa_net_bind') else None) for n, a in fake.log],
                     params=net.flat_params.clone(), buffers=net.flat_buffers.clone(), saves=saves)
    dist.barrier()
    dist.destroy_process_group()


def test_training_main_on_two_ranks_with_the_engine_stubbed_at_the_abi(tmp_path):
    """SURVEY.md section 8e on the real host code (stack_hg.main -> train -> train_step -> RMSprop.step -> validate ->
    checkpoint): per step every rank runs backward, then ONE all-reduce of the whole flat gradient, then the fused RMSprop
    with gscale = 1/world on the stream the net was bound to, then the weight re-pack; ranks draw DIFFERENT augmentations
    (seed + rank, same step counter); replicas stay bit-identical although their shard gradients differ; the logged meters
    are all-reduced (2 x names numbers, at print time only); rank 0 alone writes checkpoints."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_main_worker, args=(2, file_init_method(), str(tmp_path), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['buffers'], r1['buffers'])
    for r in (r0, r1):
        names = [n for n, _ in r['log']]
        steps = [i for i, n in enumerate(names) if n == 'pa_hg_backward']
        assert len(steps) == 4                                            # 4 synthetic batches, one epoch
        for k, i in enumerate(steps):
            end = steps[k + 1] if k + 1 < len(steps) else len(names)
            seg = names[i:end]
            ar = [j for j, n in enumerate(seg) if n == 'all_reduce' and r['log'][i + j][1][0] == 40]
            rp = seg.index('pa_rmsprop_step')
            assert len(ar) == 1 and 0 < ar[0] < rp < seg.index('pa_net_prepare_weights'), seg[:12]
            # the step's meters (stack-hg.py:176-178) read the forward pass's heat maps only: train_step asks for them between the two passes,
            # on the engine's meter stream (pa_net_meters_async on ... off), so they run beside the backward pass
            fwd = max(j for j in range(i) if names[j] == 'pa_hg_forward')
            between = names[fwd:i]
            on = [j for j, n in enumerate(between) if n == 'pa_net_meters_async']
            assert len(on) == 2, between
            assert on[0] < between.index('pa_hg_accuracy') < between.index('pa_hg_pckh') < on[1], between
        bind_stream = [a for n, a in r['log'] if n == 'pa_net_bind'][0][1]
        for n, a in r['log']:
            if n == 'pa_rmsprop_step':
                assert a[0] == 40 and a[4] == 0.5 and a[5] == bind_stream and abs(a[1] - 2.5e-4) < 1e-12
        meters = [a for n, a in r['log'] if n == 'all_reduce' and a[0] == 6]
        assert len(meters) == 4                                           # prints at i = 0, 2 (print_freq 2) and 3 (last) + the epoch's return value
    s0 = [a for n, a in r0['log'] if n == 'pa_sample_aug']
    s1 = [a for n, a in r1['log'] if n == 'pa_sample_aug']
    assert [a[1] for a in s0[:4]] == [1234] * 4 and [a[1] for a in s1[:4]] == [1235] * 4       # seed + rank
    assert [a[2] for a in s0[:4]] == [0, 1, 2, 3] == [a[2] for a in s1[:4]]
    assert len(r0['saves']) == 1 and r0['saves'][0].endswith('-0.pth.tar') and r1['saves'] == []
    files = os.listdir(os.path.join(str(tmp_path), 'dp'))
    assert any(f.endswith('-0.pth.tar') for f in files) and any(f.endswith('-0-preds.mat') for f in files) and 'train-log.txt' in files
