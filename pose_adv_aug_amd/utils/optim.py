"""RMSprop on the engine's flat parameter buffer (reference: torch.optim.RMSprop(net.parameters(),
lr, alpha=0.99, eps=1e-8, momentum=0, weight_decay=0), stack-hg.py:51-52) with the data-parallel
gradient exchange folded in: ONE all-reduce of the flat gradient over RCCL per step (replaces
nn.DataParallel's broadcast/gather/reduce_add, stack-hg.py:49)."""
import os

import torch
import torch.distributed as dist

from .. import _lib
from .._lib import lib, check, ptr, stream


def _dist_on():
    """Gradients go through the process group: world > 1, or a single rank forced through it (POSEADV_FORCE_DIST=1: RCCL
    on a one-GPU box, tests/test_gpu_rccl.py)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('POSEADV_FORCE_DIST') == '1')


class RMSprop(object):
    def __init__(self, net, lr=2.5e-4, alpha=0.99, eps=1e-8, momentum=0, weight_decay=0, overlap=False):
        """overlap (pose net, world > 1 only): exchange the gradients of a stack's hourglass (one contiguous 42 % of a 2-stack
        net) on a communication stream as soon as that stack's backward pass is enqueued, while the earlier stacks and the stem
        still run; the rest follows at the end.  Same sums, same result as the single all-reduce (off by default)."""
        if momentum != 0 or weight_decay != 0:
            raise ValueError('the reference uses momentum=0, weight_decay=0')
        self.net = net
        self._works, self._comm, self._ranges = [], None, {}
        if overlap and hasattr(net, 'num_stacks'):
            net.on_stack_done = self._exchange_stack
        net._ensure_table()
        self.param_groups = [{'lr': lr, 'alpha': alpha, 'eps': eps, 'momentum': 0, 'weight_decay': 0, 'centered': False}]
        self.square_avg = torch.zeros_like(net.flat_params)
        # fp16 build: {this gradient is non-finite, steps skipped so far} of THIS optimizer (the pose net's and the agent's step on
        # different streams in the joint stage; a process-wide flag would let one see the other's overflow)
        self._skip_state = torch.zeros(2, dtype=torch.int32, device=net.flat_params.device)
        self.steps = 0

    def zero_grad(self):
        self.net.zero_grad()

    def allreduce_grads(self):
        """Sum the flat gradient over all ranks (RCCL when the process group backend is nccl)."""
        if _dist_on():
            dist.all_reduce(self.net.flat_grads, op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0

    def _exchange_stack(self, stack):
        """hook of HourglassNet.loss_and_backward: the backward pass of `stack` has just been enqueued"""
        if not _dist_on():
            return
        import ctypes as C
        net = self.net
        h = net._net(net._last_B)
        if stack not in self._ranges:
            lo, hi = C.c_size_t(), C.c_size_t()
            check(lib().pa_hg_bucket_range(h, stack, C.byref(lo), C.byref(hi)), 'pa_hg_bucket_range')
            self._ranges[stack] = (lo.value, hi.value)
        if self._comm is None:
            self._comm = torch.cuda.Stream()
            net.flat_grads.record_stream(self._comm)
        rc = lib().pa_hg_bucket_wait(h, stack, C.c_void_p(self._comm.cuda_stream))
        if rc == -1:
            return                                   # this stream mode finishes gradients at the end only
        check(rc, 'pa_hg_bucket_wait')
        lo, hi = self._ranges[stack]
        with torch.cuda.stream(self._comm):          # the collective is ordered behind the bucket event on this stream
            work = dist.all_reduce(net.flat_grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
        self._works.append((lo, hi, work))

    def _finish_exchange(self):
        """the ranges no bucket covered, then wait for the buckets in flight"""
        n = self.net.flat_grads.numel()
        done = sorted((lo, hi) for lo, hi, _ in self._works)
        pos = 0
        for lo, hi in done + [(n, n)]:
            if lo > pos:
                dist.all_reduce(self.net.flat_grads[pos:lo], op=dist.ReduceOp.SUM)
            pos = max(pos, hi)
        for _, _, work in self._works:
            work.wait()                              # the current stream waits for the collective
        self._works = []
        return 1.0 / dist.get_world_size()

    def step(self):
        g = self.param_groups[0]
        world_scale = self._finish_exchange() if self._works else self.allreduce_grads()
        gscale = world_scale / _lib.grad_scale()                    # 1/world, and the fp16 build's gradient scale divided out
        n = self.net.flat_params.numel()
        check(lib().pa_rmsprop_step_state(ptr(self.net.flat_params), ptr(self.net.flat_grads), ptr(self.square_avg), n,
                                          float(g['lr']), float(g['alpha']), float(g['eps']), float(gscale), ptr(self._skip_state), stream()),
              'pa_rmsprop_step_state')
        self.steps += 1
        self.net.weights_changed()
        self.net._net(self.net._last_B or self.net.default_batch)      # refresh the bf16 weight copies now

    def skipped_steps(self):
        """fp16 build: optimizer steps the engine skipped because the (scaled) gradient held inf / NaN; 0 in the bf16 build"""
        import ctypes as C
        n = C.c_longlong(0)
        check(lib().pa_rmsprop_skipped_steps_state(ptr(self._skip_state), C.byref(n), stream()), 'pa_rmsprop_skipped_steps_state')
        return int(n.value)

    # torch.optim-compatible (de)serialisation: per-parameter state keyed by index, in parameters() order
    def state_dict(self):
        state = {}
        for i, (name, shape, off, numel, kind) in enumerate(t for t in self.net._table if t[4] == 0):
            state[i] = {'step': self.steps, 'square_avg': self.square_avg[off:off + numel].view(shape).clone()}
        pg = dict(self.param_groups[0])
        pg['params'] = list(range(len(state)))
        return {'state': state, 'param_groups': [pg]}

    def load_state_dict(self, sd):
        """Accepts this class's own files and torch.optim.RMSprop's: `state` is keyed by the entries of
        param_groups[0]['params'] (positions 0..n-1 in current torch, object ids in the reference's torch 0.3), in
        parameters() order either way.  A missing or mis-shaped entry is reported, never silently dropped."""
        import warnings
        params = [t for t in self.net._table if t[4] == 0]
        state = sd.get('state', {})
        keys = list(sd['param_groups'][0].get('params', range(len(params))))
        if len(keys) != len(params):
            warnings.warn('optimizer state has %d parameters, the network %d: square_avg not loaded' % (len(keys), len(params)))
            keys = []
        missing = 0
        for key, (name, shape, off, numel, kind) in zip(keys, params):
            st = state.get(key)
            if st is None or 'square_avg' not in st:
                missing += 1
                continue
            v = torch.as_tensor(st['square_avg'])
            if v.numel() != numel:
                warnings.warn('optimizer state of %s has %d elements, expected %d: skipped' % (name, v.numel(), numel))
                missing += 1
                continue
            self.square_avg[off:off + numel] = v.reshape(-1).to(self.square_avg.device, torch.float32)
            self.steps = int(st.get('step', self.steps))
        if missing and state:
            warnings.warn('optimizer state: %d of %d parameters had no square_avg entry (left at their current value)' % (missing, len(params)))
        for k, v in sd['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
