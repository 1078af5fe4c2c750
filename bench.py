#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full stage-1 training step of the 2-stack hourglass
(BASELINE.json configs[1]: bs = 24 per GPU, 256x256 MPII-shape synthetic input, bf16 storage; configs[4], the deep-stack
stress case: --stacks 8 --res 384 --bs 16 --dtype fp16):

    on-device augmentation law + bilinear warp  ->  forward  ->  Gaussian-target MSE  ->  hand-written
    backward  ->  ONE all-reduce of the flat gradient (RCCL, N > 1)  ->  fused RMSprop + bf16 weight
    re-pack  ->  PCKh (heat-map space and original resolution) on the device

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, or plain -- bench.py then
                                                          starts its N ranks itself, self_launch())

Prints ONE JSON line on rank 0 (contract in the task description).  `roofline` comes from HIP events
recorded around every launch of the MFMA kernels on the engine's own stream during a second pass of K
steps right after the timed region (same inputs, same kernels; the events are kept out of the timed
region so that they do not perturb `value`, and that pass runs with the engine's side streams off so that
an event interval contains exactly one kernel).  `cpu_baseline` is the CPU oracle (PyTorch fp32
restatement of stack-hg.py:153-180) timed on this box's host cores on a bounded sample."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# classes 0-5: maps of >= 16384 pixels (32 x 32 and larger at batch 24); the last three: every convolution launch of the smaller maps
# (16 x 16 ... 4 x 4: latency-bound launches)
PROF_NAMES = ['conv_fwd_1x1', 'conv_fwd_3x3', 'conv_dgrad_1x1', 'conv_dgrad_3x3', 'conv_wgrad_1x1', 'conv_wgrad_3x3',
              'stem_fwd_7x7', 'stem_wgrad_7x7', 'unused', 'conv_fwd_lowres', 'conv_dgrad_lowres', 'conv_wgrad_lowres']
HBM_PEAK = 8.0e12          # B/s   (MI355X_MICROARCH.md)
MFMA_PEAK = 2.5e15         # FLOP/s dense bf16


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _time_oracle(stacks, chan, B, res, steps, budget_s):
    """`steps` timed train steps (after 1 warm-up) of the CPU oracle: forward, sum-over-stacks MSE, backward, RMSprop,
    heat-map PCK -- the scope of stack-hg.py:153-180.  The batch shrinks (never the network) if one step would blow the budget."""
    from oracle import model as om, step as ostep
    from tests import inputs
    net = om.create_hg(stacks, 1, 16, chan)
    om.deterministic_fill_(net, seed=0)
    opt = ostep.make_optimizer(net)

    def batch(b):
        return (torch.from_numpy(inputs.images(1, b, res)),
                torch.from_numpy(inputs.heatmaps_from_pts(inputs.heat_pts(2, b, res=res // 4), res=res // 4)))
    img, heat = batch(2)
    ostep.pose_train_step(net, opt, img, heat)            # warm-up (allocations, oneDNN primitive caches)
    t0 = time.time()
    ostep.pose_train_step(net, opt, img, heat)
    per_img = (time.time() - t0) / 2
    if per_img * B * steps > budget_s:                    # bounded sample: shrink the batch, keep the workload
        B = max(2, int(budget_s / steps / per_img))
    img, heat = batch(B)
    if B != 2:
        ostep.pose_train_step(net, opt, img, heat)
    t1 = time.time()
    for _ in range(steps):
        ostep.pose_train_step(net, opt, img, heat)
    dt = (time.time() - t1) / steps
    return B / dt, B, dt


def cpu_baseline(stacks, chan, B, res, budget_s=40.0, steps=5):
    """The CPU oracle timed on this box's host cores: the benchmark's own configuration and the
    reference's CPU-runnable plumbing case C1 (1-stack, B = 2), >= 5 timed steps each after a warm-up."""
    # PyTorch's CPU convolutions stop scaling (and collapse when oversubscribed) well before the core count of a GPU
    # host: the thread count is the best point of the committed sweep (profiles/cpu_thread_sweep.json, tools/cpu_thread_sweep.py,
    # same host class), 32 without one; both the threads used and the cores present are reported.
    threads = 32
    try:
        sw = json.load(open(os.path.join(ROOT, 'profiles', 'cpu_thread_sweep.json')))
        if sw.get('host_cores') == os.cpu_count():
            threads = int(sw['best_threads'])
    except (OSError, ValueError, KeyError):
        pass
    threads = min(threads, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    v, b, dt = _time_oracle(stacks, chan, B, res, steps, budget_s)
    v1, b1, dt1 = _time_oracle(1, chan, 2, res, steps, 10.0)
    return {'value': v, 'unit': 'images/sec', 'cores': threads, 'host_cores': os.cpu_count(), 'cpu_model': cpu_model(), 'kind': 'port',
            'sample': '%d timed steps after a warm-up of the fp32 PyTorch-CPU oracle (oracle/step.py: forward, MSE, backward, RMSprop, '
                      'heat-map PCK), %d-stack chan %d, B=%d, %dx%d, %.2f s/step' % (steps, stacks, chan, b, res, res, dt),
            'c1': {'value': v1, 'unit': 'images/sec',
                   'sample': 'BASELINE configs[0]: 1-stack chan %d, B=%d, %dx%d, %d timed steps, %.3f s/step' % (chan, b1, res, res, steps, dt1)}}


def pckh_parity(net, aug, batch, B, res):
    """The "PCKh match" half of BASELINE.json's metric, on the driver's line (outside the timed region; the oracle is the
    CHECKER here): the engine's in-step metrics (pa_hg_accuracy, pa_hg_pckh) of one more training-mode forward pass against the oracle's
    pylib/Evaluation.py:54-97 restatement on the engine's own heat maps, and -- an untrained net on noise frames scores ~0 --
    the same two entry points on target + noise maps (pylib.Evaluation = the C ABI's pa_accuracy / pa_accuracy_origin_res)."""
    import numpy as np
    from oracle import pylib as opl
    from tests import inputs
    from pose_adv_aug_amd import pylib
    from pose_adv_aug_amd.stack_hg import PCK_IDX
    H = res // 4
    data = aug.regular(batch)
    net.forward(img4=data['img4'], pts=data['pts'])            # forward only: rank 0 runs this alone, a backward pass may hold collectives (--overlap)
    maps = net.heatmaps(B)[-1].cpu()
    target = pylib.HumanPts.pts2heatmap_batch(data['pts'], H, H).cpu()
    c, s, r = data['c'].cpu().float(), data['s'].cpu().float().view(B, 1), data['r'].cpu().float().view(B, 1)
    g, nm = data['grnd_pts'].cpu().float(), data['normalizer'].cpu().float()
    e_acc = net.accuracy(PCK_IDX).cpu().numpy()
    e_pck = net.pckh_origin_res(data['c'], data['s'], data['r'], data['grnd_pts'], data['normalizer'])[0].cpu().numpy()
    o_acc = opl.accuracy(maps, target, PCK_IDX).numpy()
    o_pck = opl.accuracy_origin_res(maps, c, s, [H, H], g, nm, r).numpy()
    noisy = torch.from_numpy(inputs.noisy_heatmaps(43, target.numpy(), noise=0.2))
    e2_acc = pylib.Evaluation.accuracy(noisy, target, PCK_IDX).cpu().numpy()
    e2_pck = pylib.Evaluation.accuracy_origin_res(noisy, c, s, [H, H], g, nm, r).cpu().numpy()
    o2_acc = opl.accuracy(noisy, target, PCK_IDX).numpy()
    o2_pck = opl.accuracy_origin_res(noisy, c, s, [H, H], g, nm, r).numpy()
    diffs = [np.abs(a - b).max() for a, b in ((e_acc, o_acc), (e_pck, o_pck), (e2_acc, o2_acc), (e2_pck, o2_pck))]
    return {'tolerance': 1e-4, 'abs_diff': float(max(diffs)), 'match': bool(max(diffs) <= 1e-4),
            'untrained_net': {'engine': {'pck_heatmap': float(e_acc[0]), 'pckh_origin_res': float(e_pck[0])},
                              'oracle': {'pck_heatmap': float(o_acc[0]), 'pckh_origin_res': float(o_pck[0])}},
            'target_plus_noise_maps': {'engine': {'pck_heatmap': float(e2_acc[0]), 'pckh_origin_res': float(e2_pck[0])},
                                       'oracle': {'pck_heatmap': float(o2_acc[0]), 'pckh_origin_res': float(o2_pck[0])}},
            'per_joint_values_compared': int(sum(len(x) for x in (e_acc, e_pck, e2_acc, e2_pck)))}


def cold_rate(dev, mb=(25, 50, 100, 400)):
    """What "just moving the bytes" reaches on this box, with the ENGINE'S OWN plain streaming kernel (pa_copy_probe: 16 B per lane,
    elementwise.hip) -- not a framework kernel: a copy of a buffer of each size that has not been touched for 640 MB of other traffic
    (the step's tensors are cold: ~19 GB move between two uses), one HIP-event pair per launch, median of 7; and the large-buffer rate
    (1.6 GB copy, warm-up + median of 5) that a launch's ramp and tail no longer dent.  Returns (rate at the median per-launch size,
    per-size table, large-buffer rate), TB/s of read + written bytes."""
    from pose_adv_aug_amd import _lib
    lib, ptr, check, stream = _lib.lib(), _lib.ptr, _lib.check, _lib.stream
    flush_a = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
    flush_b = torch.empty(320 << 20, dtype=torch.uint8, device=dev)

    def timed(dst, src, nbytes, flush):
        if flush:
            check(lib.pa_copy_probe(ptr(flush_b), ptr(flush_a), flush_a.numel(), stream()), 'pa_copy_probe')
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(lib.pa_copy_probe(ptr(dst), ptr(src), nbytes, stream()), 'pa_copy_probe'); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3
    out = {}
    for m in mb:
        n = m << 20
        a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
        b = torch.empty(n, dtype=torch.uint8, device=dev)
        ts = sorted(timed(b, a, n, True) for _ in range(7))
        out[m] = 2.0 * n / ts[len(ts) // 2] / 1e12
        del a, b
    n = 1600 << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    timed(b, a, n, False)
    ts = sorted(timed(b, a, n, False) for _ in range(5))
    large = 2.0 * n / ts[len(ts) // 2] / 1e12
    # which COPY FORM reaches what on this box (the guide quotes 6.29 TB/s for a "float4 copy"): the same 1.6 GB in the probe's three forms
    forms = {}
    for form, name in ((0, 'grid_stride_4_in_flight'), (1, 'one_16B_chunk_per_thread'), (2, 'grid_stride_nontemporal')):
        def tf():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); check(lib.pa_copy_probe_form(ptr(b), ptr(a), n, form, stream()), 'pa_copy_probe_form'); e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3
        tf()
        ts = sorted(tf() for _ in range(5))
        forms[name] = round(2.0 * n / ts[len(ts) // 2] / 1e12, 2)
    # ... and on 256 MB, which the 256 MB Infinity Cache partly holds between repetitions (a small-buffer copy loop reads more than HBM gives)
    n2 = 256 << 20
    ts = sorted(timed(b[:n2], a[:n2], n2, False) for _ in range(5))
    forms['grid_stride_4_in_flight_256MB_warm'] = round(2.0 * n2 / ts[len(ts) // 2] / 1e12, 2)
    cold_rate.forms = forms
    # the large-buffer rate of the floor = the best form's (round 6: the grid-stride form of rounds 4 - 5 reaches 4.3 - 4.6 TB/s on boxes where
    # the one-chunk-per-thread form reaches 6.2, the guide's figure -- the 4.58 of round 5 was the probe's form, not the box)
    large = max(large, forms['one_16B_chunk_per_thread'], forms['grid_stride_nontemporal'])
    sizes = sorted(k for k in out if k <= 100)
    return out[sizes[len(sizes) // 2]], {('%dMB' % k): round(v, 2) for k, v in out.items()}, large


def design_floor(net, opt, B, res, h, frame_hw=(720, 1280)):
    """`roofline.floor`: the bytes THIS fused design moves per step, all from the library's own tables -- the engine's launch table
    (pa_net_design_bytes: activations, both operands of the BatchNorm-backward loads, reference tensors, stored dz tensors, fp32 slabs
    written + read back, weights), the crop's stage buffers (pa_crop_design_bytes: an upper bound, worst-case windows), RMSprop
    (p, g, v read; p, v written) and the weight re-pack (fp32 read, two 16-bit copies written) from the parameter count -- at two
    rates measured in this run with the engine's own copy kernel: cold at the step's per-launch tensor sizes (what one launch of
    25-100 MB can see) and the large-buffer rate (what the memory system sustains).  Algorithmic bytes beside it."""
    from pose_adv_aug_amd import _lib
    rw = (C.c_double * 2)()
    _lib.check(_lib.lib().pa_net_design_bytes(h, rw))
    crop = (C.c_double * 2)()
    _lib.check(_lib.lib().pa_crop_design_bytes(B, frame_hw[0], frame_hw[1], res, crop))
    npar = int(net.flat_params.numel())
    opt_rd, opt_wr = 3 * 4.0 * npar + 4.0 * npar, 2 * 4.0 * npar + 2 * 2.0 * npar
    total = rw[0] + rw[1] + crop[0] + crop[1] + opt_rd + opt_wr
    rate, table, large = cold_rate(net.flat_params.device)
    return {'design_bytes_per_step': round(total), 'design_read': round(rw[0] + crop[0] + opt_rd), 'design_write': round(rw[1] + crop[1] + opt_wr),
            'parts': {'net_read': round(rw[0]), 'net_write': round(rw[1]), 'crop_read_bound': round(crop[0]), 'crop_write_bound': round(crop[1]),
                      'optimizer_repack_read': round(opt_rd), 'optimizer_repack_write': round(opt_wr)},
            'algorithmic_bytes_per_step': round(380.5e6 * B) if res == 256 else None,
            'cold_rate_TBps': round(rate, 2), 'cold_rate_table_TBps': table, 'rate_large_TBps': round(large, 2),
            'rate_large_by_copy_form_TBps': getattr(cold_rate, 'forms', None),
            'rate_large_note': 'rate_large_TBps = 1.6 GB device-to-device copy (read + written bytes), median of 5, the BEST of pa_copy_probe\'s three forms on this box at its own clocks; '
                               'rate_large_by_copy_form_TBps lists the other forms of the same kernel (the guide\'s 6.29 TB/s is a float4 copy = form one_16B_chunk_per_thread) '
                               'and a 256 MB copy, which the 256 MB Infinity Cache partly serves',
            'floor_ms': round(total / (rate * 1e12) * 1e3, 3), 'floor_ms_at_large_rate': round(total / (large * 1e12) * 1e3, 3),
            'floor_ms_at_8TBps': round(total / 8e12 * 1e3, 3),
            'algorithmic_floor_ms_at_large_rate': round(380.5e6 * B / (large * 1e12) * 1e3, 3) if res == 256 else None,
            'note': 'rates: pa_copy_probe (the engine\'s own 16 B/lane copy) measured in this run; floor_ms = design bytes / cold rate at the '
                    'median per-launch size (25-100 MB buffers untouched for 640 MB of traffic); floor_ms_at_large_rate = design bytes / 1.6 GB copy rate'}


def self_launch(n):
    """Start ranks 0..n-1 of this command (same argv) and wait for them: RANK / LOCAL_RANK / WORLD_SIZE in the environment,
    rendezvous through a file store (POSEADV_DIST_INIT=file://...), HSA_ENABLE_IPC_MODE_LEGACY=0 kept for RCCL's dmabuf IPC.
    The children inherit stdout / stderr, so rank 0's JSON line is this command's output.  Returns the worst exit code; a rank
    that dies takes the others down instead of leaving them in a collective."""
    import subprocess
    import tempfile
    store = tempfile.NamedTemporaryFile(prefix='poseadv_bench_store_', delete=False)
    store.close()
    os.unlink(store.name)                                    # FileStore creates it
    env = dict(os.environ, WORLD_SIZE=str(n), POSEADV_DIST_INIT='file://' + store.name, POSEADV_SELF_LAUNCHED='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    code = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:                        # exact PIDs we started, nothing else
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink(store.name)
        except OSError:
            pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed steps (SURVEY.md section 8d protocol: 20 warm-up + 100 timed)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--bs', type=int, default=24, help='per-GPU batch (BASELINE config: 24)')
    ap.add_argument('--stacks', type=int, default=2)
    ap.add_argument('--chan', type=int, default=256)
    ap.add_argument('--res', type=int, default=256, help='network input resolution (SURVEY.md C5: --stacks 8 --res 384 --bs 16)')
    ap.add_argument('--dtype', choices=['bf16', 'fp16'], default='bf16', help='16-bit storage / MFMA operand type (BASELINE configs[4]: fp16)')
    ap.add_argument('--overlap', type=int, default=0, help='1 (N > 1): exchange each stack\'s hourglass gradients during the rest of the backward pass (RMSprop(overlap=True))')
    ap.add_argument('--fin-rows', type=int, default=-1, help='>= 0: pa_net_set_fin_prologue(rows) -- row limit of the BatchNorm finalize in the consumer\'s prologue, 0 = every finalize a launch (A/B; default: the library\'s 128)')
    ap.add_argument('--graph', type=int, default=0, help='1: forward + backward replayed from a captured HIP graph (pa_hg_train_step)')
    ap.add_argument('--single-stream', type=int, default=0, help='1: the engine\'s side / weight-gradient streams off for the timed region too (experiments)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-traffic', action='store_true', help='do not start the rocprofv3 child passes (PMC HBM traffic, kernel trace) of this command; '
                    'by default they run at N = 1 when rocprofv3 is on PATH, and the recorded profiles/ numbers are quoted otherwise')
    ap.add_argument('--no-floor', action='store_true', help='skip the design-bytes / cold-rate floor of the roofline object')
    ap.add_argument('--keep-profiles', action='store_true', help='keep the child passes\' rocprofv3 output under gpurun_out/')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, FileStore rendezvous -- no TCP port),
        # exactly what `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` starts; rank 0 prints the line
        raise SystemExit(self_launch(args.gpus))

    from pose_adv_aug_amd import _lib
    _lib.set_dtype(args.dtype)
    from pose_adv_aug_amd.stack_hg import init_distributed, broadcast_parameters, train_step
    from pose_adv_aug_amd.data import AugmentAhead, Augmenter, DeviceBatch
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop

    _lib.require_gpu()
    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d needs WORLD_SIZE=%d (launch N > 1 with torch.distributed.run)' % (args.gpus, args.gpus))
    dev = torch.device('cuda', torch.cuda.current_device())
    B, res = args.bs, args.res
    net = create_hg(args.stacks, 1, 16, args.chan, res=res, default_batch=B)
    net.reset_parameters(seed=0)
    net.use_graph = bool(args.graph)
    broadcast_parameters(net)
    opt = RMSprop(net, lr=2.5e-4, alpha=0.99, eps=1e-8, overlap=bool(args.overlap) and (world > 1 or os.environ.get('POSEADV_FORCE_DIST') == '1'))
    aug = Augmenter(seed=100 + rank, inp_res=res, out_res=res // 4)
    batches = [DeviceBatch.synthetic(B, seed=rank * 100 + k) for k in range(2)]       # resident in HBM
    net.train()
    if args.single_stream:
        _lib.check(_lib.lib().pa_net_set_multi_stream(net._net(B), 0))
    if args.fin_rows >= 0:
        _lib.check(_lib.lib().pa_net_set_fin_prologue(net._net(B), args.fin_rows))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ahead = AugmentAhead(aug)

    def run(n, mark=None):
        """n steps of stack_hg.train's loop body: the NEXT batch's augmentation (law + crop + joints, all inside the timed
        region) is enqueued one step ahead on its own stream, like the reference's DataLoader workers"""
        out = None
        ahead.start(batches[0])
        for i in range(n):
            data = ahead.take()
            ahead.start(batches[(i + 1) % len(batches)] if i + 1 < n else None)
            out = train_step(net, opt, aug, batches[i % len(batches)], data=data)
            if mark is not None:
                mark(i)
        return out

    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    loss, pckh, pckh_o = run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
    value = world * B * args.steps / dt
    # who actually took part: a one per rank summed THROUGH the collective backend, and the distinct devices behind the ranks
    ranks_seen, devices_seen, backend = 1, 1, None
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones[0])))
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, '%s:%s' % (os.uname().nodename, getattr(torch.cuda.get_device_properties(dev), 'uuid', torch.cuda.current_device())))
        devices_seen, backend = len(set(ids)), dist.get_backend()

    # median of per-step event intervals, in a pass of its own: an event record between two kernels of a queue
    # costs that queue a bubble, so it stays out of the timed region that `value` comes from
    median_ms = None
    if os.environ.get('PA_BENCH_CHILD') != '1':          # (not inside the rocprofv3 child passes: their step counts are fixed)
        nm = min(args.steps, 50)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nm + 1)]
        evs[0].record()
        run(nm, mark=lambda i: evs[i + 1].record())
        sync()
        per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nm))
        median_ms = per_step[nm // 2]

    roofline = None
    if not args.no_roofline:
        h = net._net(B)
        # per-kernel durations are taken with the engine's side streams off: with them on, launches of independent
        # branches overlap and a launch's event interval contains other kernels' work
        net.use_graph = False
        _lib.check(_lib.lib().pa_net_set_multi_stream(h, 0))
        _lib.check(_lib.lib().pa_net_profile_begin(h))
        run(args.steps)
        rep = (C.c_double * 64)()
        _lib.check(_lib.lib().pa_net_profile_report(h, rep, 16, None))
        _lib.check(_lib.lib().pa_net_set_multi_stream(h, 1))
        if rank == 0 and os.environ.get('PA_BENCH_SEQ_OUT'):            # for tools/trace_classes.py (rocprofv3 runs of this command)
            n = _lib.lib().pa_net_profile_classes(h, None, 0)
            seq = (C.c_int32 * n)()
            _lib.lib().pa_net_profile_classes(h, seq, n)
            json.dump({'classes': PROF_NAMES, 'sequence': list(seq), 'steps': args.steps}, open(os.environ['PA_BENCH_SEQ_OUT'], 'w'))
        rows = []
        for i, name in enumerate(PROF_NAMES):
            ms, cnt, by, fl = rep[4 * i], rep[4 * i + 1], rep[4 * i + 2], rep[4 * i + 3]
            if cnt > 0:
                rows.append({'kernel': name, 'ms_total': ms, 'launches': int(cnt), 'avg_us': 1e3 * ms / cnt,
                             'bytes': by, 'flops': fl})
        dom = max(rows, key=lambda r: r['ms_total'])
        ai = dom['flops'] / dom['bytes']
        bound = 'mfma' if ai > MFMA_PEAK / HBM_PEAK else 'hbm'
        secs = dom['ms_total'] * 1e-3
        if bound == 'mfma':
            ach, peak, unit = dom['flops'] / secs / 1e12, MFMA_PEAK / 1e12, 'TFLOP/s'
        else:
            ach, peak, unit = dom['bytes'] / secs / 1e9, HBM_PEAK / 1e9, 'GB/s'
        conv_ms = sum(r['ms_total'] for r in rows) / args.steps
        # HBM bytes per launch of that kernel class from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc
        # passes) and its trace-grade duration (rocprofv3 --kernel-trace matched launch by launch with the engine's class sequence):
        # MEASURED IN THIS RUN by child passes of this very command (tools/inrun_prof.py) when rocprofv3 is on PATH (rank 0, N = 1);
        # otherwise the numbers committed under profiles/ are quoted, tagged as recorded.
        traffic, trace = None, None
        is_c2 = (args.stacks, args.chan, res, B, args.dtype) == (2, 256, 256, 24, 'bf16')
        wl = ['--bs', str(B), '--stacks', str(args.stacks), '--chan', str(args.chan), '--res', str(res), '--dtype', args.dtype]
        measured_pmc = measured_tr = None
        if rank == 0 and world == 1 and not args.no_traffic:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import inrun_prof
            torch.cuda.synchronize()
            measured_pmc = inrun_prof.measure_traffic(wl, keep=args.keep_profiles)
            measured_tr = inrun_prof.measure_trace(wl, keep=args.keep_profiles)
        if measured_pmc and dom['kernel'] in measured_pmc['classes']:
            c = measured_pmc['classes'][dom['kernel']]
            traffic = {'hbm_bytes_per_launch': round(c['hbm_bytes_per_launch'], 1), 'launches_profiled': c['launches_profiled'],
                       'whole_step': {'fetch_bytes_x2': round(measured_pmc['fetch_bytes_per_step_x2']), 'write_bytes': round(measured_pmc['write_bytes_per_step']),
                                      'excluded_one_time_workspace_fill_bytes': round(measured_pmc.get('excluded_one_time_fill_bytes', 0.0))},
                       'source': 'measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) and --pmc WRITE_SIZE child passes of this command, 7 steps each; per class by launch order'}
        else:
            for name in ('round6_pmc.json', 'round5_pmc.json', 'round4_pmc.json', 'round3_pmc.json', 'round2_pmc.json'):
                pmc_path = os.path.join(ROOT, 'profiles', name)
                if is_c2 and os.path.isfile(pmc_path):
                    pmc = json.load(open(pmc_path))
                    if dom['kernel'] in pmc:
                        traffic = {'hbm_bytes_per_launch': round(pmc[dom['kernel']]['hbm_bytes_per_launch'], 1),
                                   'source': 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this command, recorded; not measured in this run)' % name}
                        break
        tr, tr_src = None, None
        if measured_tr and dom['kernel'] in measured_tr:
            tr, tr_src = measured_tr[dom['kernel']], 'measured in this run: rocprofv3 --kernel-trace child pass of this command (single-stream roofline leg, 5 steps)'
        else:
            for name in ('round6_trace_classes.json', 'round5_trace_classes.json', 'round4_trace_classes.json', 'round3_trace_classes.json', 'round2_trace_classes.json'):
                tr_path = os.path.join(ROOT, 'profiles', name)
                if is_c2 and os.path.isfile(tr_path):
                    tr, tr_src = json.load(open(tr_path)).get(dom['kernel']), 'profiles/%s (recorded; not measured in this run)' % name
                    break
        if tr:
            t_ach = (dom['flops'] if bound == 'mfma' else dom['bytes']) / dom['launches'] / (tr['avg_us'] * 1e-6) / (1e12 if bound == 'mfma' else 1e9)
            trace = {'avg_kernel_us': tr['avg_us'], 'launches_per_step': tr['launches_per_step'], 'achieved': round(t_ach, 2),
                     'frac': round(t_ach / peak, 4), 'source': tr_src}
            if measured_tr:
                trace['classes_us'] = {k: v['avg_us'] for k, v in measured_tr.items()}
        floor = None
        if not args.no_floor:
            floor = design_floor(net, opt, B, res, h)
        # the CRITICAL-PATH class beside the dominant-by-time one: the main chain's largest class (the 1x1 data gradients run on the caller's
        # stream, one after the other; the weight-gradient groups that dominate by time run beside them on a queue of their own)
        main_chain = None
        mc = [r for r in rows if r['kernel'] == 'conv_dgrad_1x1']
        if mc:
            r = mc[0]
            mc_ach = r['bytes'] / (r['ms_total'] * 1e-3) / 1e9
            main_chain = {'kernel': r['kernel'], 'bound': 'hbm', 'avg_us': round(r['avg_us'], 2), 'launches_per_step': r['launches'] // args.steps,
                          'alg_bytes_per_launch': r['bytes'] / r['launches'], 'achieved': round(mc_ach, 2), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                          'frac': round(mc_ach / (HBM_PEAK / 1e9), 4), 'traffic': None, 'trace': None}
            if measured_pmc and r['kernel'] in measured_pmc['classes']:
                main_chain['traffic'] = {'hbm_bytes_per_launch': round(measured_pmc['classes'][r['kernel']]['hbm_bytes_per_launch'], 1),
                                         'source': 'measured in this run (the same rocprofv3 --pmc child passes as roofline.traffic)'}
            if measured_tr and r['kernel'] in measured_tr:
                tu = measured_tr[r['kernel']]['avg_us']
                main_chain['trace'] = {'avg_kernel_us': tu, 'frac': round(r['bytes'] / r['launches'] / (tu * 1e-6) / HBM_PEAK, 4)}
        roofline = {'bound': bound, 'achieved': round(ach, 2), 'peak': peak, 'unit': unit, 'frac': round(ach / peak, 4),
                    'traffic': traffic, 'trace': trace, 'floor': floor, 'main_chain': main_chain, 'kernel': dom['kernel'], 'avg_launch_us': round(dom['avg_us'], 2),
                    'launches_per_step': dom['launches'] // args.steps,
                    'alg_bytes_per_launch': dom['bytes'] / dom['launches'], 'alg_flops_per_launch': dom['flops'] / dom['launches'],
                    'mfma_kernels_ms_per_step': round(conv_ms, 3),
                    'note': 'kernel = the MFMA-kernel class with the largest total HIP-event time in the single-stream leg.  conv_wgrad_3x3 launches are '
                            'GROUP launches since round 5 (wgrad_group_kernel: the 3x3 + 1x1 weight gradients of a residual block, or the head\'s layers, '
                            'in one launch of <= 256 workgroups, bytes / flops of all its jobs): built to run beside the main chain on a share of the CUs, '
                            'so their time ALONE over-states what they cost the step; classes[conv_dgrad_1x1] is the main chain\'s largest class',
                    'classes': {r['kernel']: {'ms_per_step': round(r['ms_total'] / args.steps, 3),
                                              'tflops': round(r['flops'] / (r['ms_total'] * 1e-3) / 1e12, 1),
                                              'gbps': round(r['bytes'] / (r['ms_total'] * 1e-3) / 1e9, 1),
                                              'hbm_frac': round(r['bytes'] / (r['ms_total'] * 1e-3) / HBM_PEAK, 4), 'launches_per_step': r['launches'] // args.steps,
                                              'avg_us': round(r['avg_us'], 2)} for r in rows},
                    'whole_step': ({'hbm_frac': round(value / world * 380.5e6 / HBM_PEAK, 4),
                                    'mfma_frac': round(value / world * 50.0e9 / MFMA_PEAK, 4)}
                                   if (args.stacks, args.chan, res) == (2, 256, 256) else None)}

    parity = None
    if rank == 0 and not args.no_parity:
        parity = pckh_parity(net, aug, batches[0], B, res)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.stacks, args.chan, B, res)

    if rank == 0:
        line = {'metric': 'images/sec, %d-stack HG %dx%d bs=%d per GPU, full training step' % (args.stacks, res, res, B), 'value': round(value, 2),
                'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(1e3 * dt / args.steps, 3), 'ms_per_step_median': round(median_ms, 3) if median_ms is not None else None, 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
                'config': {'workload': '%s%d-stack hourglass chan %d, bs=%d/GPU, %dx%d MPII-shape synthetic frames '
                                       '(720x1280 uint8 resident in HBM), on-device HumanAug warp, heat-map MSE, RMSprop, PCKh'
                                       % ('BASELINE configs[1]: ' if (args.stacks, args.chan, res, B, args.dtype) == (2, 256, 256, 24, 'bf16') else
                                          ('BASELINE configs[4]: ' if (args.stacks, args.chan, res, B, args.dtype) == (8, 256, 384, 16, 'fp16') else ''), args.stacks, args.chan, B, res, res),
                           'global_batch': world * B, 'parallelism': 'dp%d' % world + ('+overlapped-exchange' if args.overlap and world > 1 else ''),
                           'ranks_seen': ranks_seen, 'devices_seen': devices_seen, 'collective_backend': backend,
                           'loss': float(loss), 'pckh': float(pckh), 'pckh_origin_res': float(pckh_o)},
                'pckh_parity': parity, 'roofline': roofline, 'cpu_baseline': cpu}
        print(json.dumps(line))
    if dist.is_available() and dist.is_initialized():          # (world > 1, or one rank forced through RCCL: POSEADV_FORCE_DIST=1)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
