"""models/asn_stacked_hg.py of the reference behind the HIP engine.

`create_hg` / `create_asn` keep the reference's signatures (models/asn_stacked_hg.py:344-347, :441-444)
and return objects with the nn.Module surface the reference's scripts use -- __call__/forward,
state_dict / load_state_dict (same names, NCHW shapes and order as the reference's modules),
parameters(), train(), eval(), cuda() -- but the arithmetic runs in libposeadv_hip.so on flat fp32
parameter / gradient buffers; torch only owns the memory.  There is no autograd graph: use
`loss_and_backward()` (the fused forward + loss + hand-written backward) instead of loss.backward().
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np
import torch

from .. import _lib
from .._lib import lib, check, ptr, stream, require_gpu, PoseAdvError


class _HipModule(object):
    """Flat-buffer module: parameters / gradients / running statistics live in three flat fp32 GPU
    tensors laid out by the C library (pa_net_tensor_info), state_dict entries are views into them."""

    def __init__(self):
        self.training = True
        self._nets = {}          # batch size -> (handle, workspace tensor)
        self._table = None
        self.flat_params = self.flat_grads = self.flat_buffers = None
        self._nbt = 0            # num_batches_tracked (identical for every BatchNorm of the module)
        self._weights_dirty = True

    # -- to be provided by subclasses
    def _create(self, B):
        raise NotImplementedError

    # -- handle management
    def _read_table(self, h):
        L = lib()
        table = []
        name = C.create_string_buffer(256)
        shape = (C.c_int * 4)(); nd = C.c_int(); off = C.c_size_t(); numel = C.c_size_t(); kind = C.c_int()
        for i in range(L.pa_net_num_tensors(h)):
            check(L.pa_net_tensor_info(h, i, name, 256, shape, C.byref(nd), C.byref(off), C.byref(numel), C.byref(kind)))
            table.append((name.value.decode(), tuple(shape[k] for k in range(nd.value)), off.value, numel.value, kind.value))
        return table

    def _net(self, B):
        require_gpu()
        if B not in self._nets:
            L = lib()
            h = self._create(B)
            if not h:
                raise PoseAdvError('network creation failed: %s' % L.pa_last_error().decode())
            dev = _lib.device()
            if self._table is None:
                self._table = self._read_table(h)
                self.flat_params = torch.zeros(L.pa_net_param_floats(h), dtype=torch.float32, device=dev)
                self.flat_grads = torch.zeros_like(self.flat_params)
                self.flat_buffers = torch.zeros(max(1, L.pa_net_buffer_floats(h)), dtype=torch.float32, device=dev)
                for name, shape, off, numel, kind in self._table:
                    if kind == 1 and name.endswith('running_var'):
                        self.flat_buffers[off:off + numel] = 1.0
                self.reset_parameters()
            ws = torch.empty(L.pa_net_workspace_bytes(h), dtype=torch.uint8, device=dev)       # (pa_net_bind clears it)
            check(L.pa_net_bind(h, ptr(self.flat_params), ptr(self.flat_grads), ptr(self.flat_buffers), ptr(ws), stream()),
                  'pa_net_bind')
            self._nets[B] = (h, ws)               # (binding packs THIS handle's bf16 weights; a pending change still has to reach
            if len(self._nets) == 1:              #  the handles of the other batch sizes, so the flag survives unless this is the only one)
                self._weights_dirty = False
        h, _ = self._nets[B]
        if self._weights_dirty:
            for hh, _ in self._nets.values():
                check(lib().pa_net_prepare_weights(hh), 'pa_net_prepare_weights')
            self._weights_dirty = False
        return h

    def _ensure_table(self):
        if self._table is None:
            self._net(self.default_batch)

    def __del__(self):
        try:
            for h, _ in self._nets.values():
                lib().pa_net_destroy(h)
        except Exception:
            pass

    # -- nn.Module surface
    def weights_changed(self):
        """Call after writing into parameter views (optimizer step, load_state_dict)."""
        self._weights_dirty = True

    def named_parameters(self):
        self._ensure_table()
        for name, shape, off, numel, kind in self._table:
            if kind == 0:
                yield name, self.flat_params[off:off + numel].view(shape)

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def named_grads(self):
        """views into flat_grads; they carry _lib.grad_scale() (1 for the bf16 build)"""
        self._ensure_table()
        for name, shape, off, numel, kind in self._table:
            if kind == 0:
                yield name, self.flat_grads[off:off + numel].view(shape)

    def state_dict(self, prefix=''):
        self._ensure_table()
        sd = OrderedDict()
        for name, shape, off, numel, kind in self._table:
            if kind == 0:
                sd[prefix + name] = self.flat_params[off:off + numel].view(shape)
            elif kind == 1:
                sd[prefix + name] = self.flat_buffers[off:off + numel].view(shape)
            else:
                sd[prefix + name] = torch.tensor(self._nbt, dtype=torch.long)
        return sd

    def load_state_dict(self, state_dict, strict=True):
        own = self.state_dict()
        missing = [k for k in own if k not in state_dict and 'module.' + k not in state_dict]
        unexpected = []
        for k, v in state_dict.items():
            kk = k[7:] if k.startswith('module.') else k        # reference checkpoints carry DataParallel's prefix
            if kk not in own:
                unexpected.append(k)
                continue
            if kk.endswith('num_batches_tracked'):
                self._nbt = int(v)
                continue
            own[kk].copy_(torch.as_tensor(v).to(own[kk].device, torch.float32).view_as(own[kk]))
        if strict and (missing or unexpected):
            raise KeyError('load_state_dict: missing %s unexpected %s' % (missing[:5], unexpected[:5]))
        self.weights_changed()
        return missing, unexpected

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def cuda(self, *a, **k):
        require_gpu()
        return self

    def zero_grad(self):
        if self.flat_grads is not None:
            self.flat_grads.zero_()

    def num_params(self):
        return sum(n for _, _, _, n, k in (self._table or []) if k == 0)

    def reset_parameters(self, seed=None):
        """The reference's initialisation (models/asn_stacked_hg.py:258-270): conv weight and bias
        U(+-1/sqrt(k*k*Cin)), BatchNorm gamma U(0,1), beta 0; Linear layers keep torch's default."""
        if self._table is None:
            self._net(self.default_batch)          # creates the buffers and calls back into this method
            if seed is None:
                return
        g = torch.Generator().manual_seed(int(seed) if seed is not None else torch.initial_seed() % (2 ** 31))
        bound = None
        for name, shape, off, numel, kind in self._table:
            if kind != 0:
                continue
            if len(shape) == 4:
                bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
                v = (torch.rand(shape, generator=g) * 2 - 1) * bound
            elif len(shape) == 2:                               # nn.Linear default: U(+-1/sqrt(fan_in))
                bound = 1.0 / math.sqrt(shape[1])
                v = (torch.rand(shape, generator=g) * 2 - 1) * bound
            elif _is_bn_name(name):
                v = torch.rand(shape, generator=g) if name.endswith('weight') else torch.zeros(shape)
            else:                                               # bias of the conv / linear just declared
                v = (torch.rand(shape, generator=g) * 2 - 1) * bound
            self.flat_params[off:off + numel] = v.reshape(-1).to(self.flat_params.device)
        self.weights_changed()


def _is_bn_name(name):
    leaf = name.rsplit('.', 2)
    mod = leaf[-2] if len(leaf) >= 2 else ''
    if mod.startswith('bn'):
        return True
    return mod == '1' and '.linear.' in '.' + name        # linear.<i>.1 is the BatchNorm of the Sequential


class HourglassNet(_HipModule):
    """_Hourglass_Wrapper of the reference (models/asn_stacked_hg.py:215-342)."""

    def __init__(self, num_modules, num_stacks, chan=256, num_classes=16, res=256, default_batch=24):
        super().__init__()
        if num_modules != 1:
            raise ValueError('the reference scripts always use num_modules=1; that is what the engine builds')
        self.num_stacks, self.chan, self.num_classes, self.res = num_stacks, chan, num_classes, res
        self.default_batch = default_batch
        self._last_B = None

    def _create(self, B):
        return lib().pa_hg_create(self.num_stacks, self.num_classes, self.chan, B, self.res)

    def forward(self, x=None, asn=None, is_half_hg=False, is_aug=False, is_dropout=False, img4=None, pts=None,
                dropout_masks=None, seed=0):
        """models/asn_stacked_hg.py:282-342.  x: [B][3][res][res] fp32 GPU tensor (or img4: the bf16
        NHWC4 output of the on-device warp).  Returns the list of per-stack heat maps [B][16][res/4][res/4]
        (fp32, NCHW) like the reference.  With `asn` and is_half_hg the agent's two logit tensors.
        Occlusion branch (:308-324): with an is_dropout agent, is_half_hg gives the [B][1][4][4] mask logits; the whole
        hourglass draws two cells per sample from their softmax on the device (stream: `seed` and a per-module call
        counter), zeroes them in the neck / skip tensors of every stack and returns (outs, pred_mask, indexes); the
        drawn masks stay in `last_dropout_masks` for loss_and_backward(dropout_masks=...).  `dropout_masks`
        ([B][1][4][4], without an agent) applies given masks (:183-189)."""
        if asn is not None:
            assert is_aug != is_dropout                      # :299
            if is_aug:
                return asn._forward_from_pose(self, x, img4, is_half_hg)
            assert dropout_masks is None                     # :160
            pred_mask = asn._forward_masks_from_pose(self, x, img4, update_running=is_half_hg)
            if is_half_hg:
                return pred_mask
            masks, indexes = sample_mask(pred_mask, seed=seed, step=self._drop_calls)
            self._drop_calls += 1
            self.last_dropout_masks = masks
            return self._forward(x, img4, pts, masks), pred_mask, indexes
        return self._forward(x, img4, pts, dropout_masks)

    __call__ = forward
    _drop_calls = 0
    last_dropout_masks = None

    def _set_masks(self, h, masks):
        """pa_hg_set_dropout_masks with a [B][1][4][4] (or [B][16]) mask tensor, None = off; returns the tensor to keep alive"""
        if masks is None:
            check(lib().pa_hg_set_dropout_masks(h, None), 'pa_hg_set_dropout_masks')
            return None
        m = masks.to(self.flat_params.device, torch.float32).reshape(masks.shape[0], 16).contiguous()
        check(lib().pa_hg_set_dropout_masks(h, ptr(m)), 'pa_hg_set_dropout_masks')
        return m

    def _forward(self, x, img4, pts, dropout_masks=None):
        B = x.shape[0] if x is not None else img4.shape[0]
        h = self._net(B)
        self._last_B = B
        p = pts.to(torch.float64).contiguous() if pts is not None else None
        losses = torch.empty(self.num_stacks, dtype=torch.float32, device=self.flat_params.device) if pts is not None else None   # (fully overwritten)
        keep = self._set_masks(h, dropout_masks)
        try:
            check(lib().pa_hg_forward(h, ptr(x.contiguous().float()) if x is not None else None, ptr(img4), ptr(p),
                                      1 if self.training else 0, ptr(losses)), 'pa_hg_forward')
        finally:
            if keep is not None:                             # (stream-ordered: `keep` outlives the kernels that read it)
                self._set_masks(h, None)
        if self.training:
            self._nbt += 1
        self._last_losses = losses
        return self.heatmaps(B)

    def heatmaps(self, B=None):
        B = B or self._last_B
        h = self._net(B)
        outs = []
        for i in range(self.num_stacks):
            o = torch.empty((B, 16, self.res // 4, self.res // 4), dtype=torch.float32, device=self.flat_params.device)
            check(lib().pa_hg_heatmap_nchw(h, i, ptr(o)), 'pa_hg_heatmap_nchw')
            outs.append(o)
        return outs

    use_graph = False        # loss_and_backward(img4=...) replays a captured HIP graph of forward + backward (pa_hg_train_step)
    on_stack_done = None     # callback(stack index) after the backward pass of a stack is enqueued (utils.optim.RMSprop(overlap=True))

    def loss_and_backward(self, x=None, pts=None, img4=None, want_outputs=False, dropout_masks=None, after_forward=None):
        """One pass of stack-hg.py:153-164 without the optimizer: forward in the current mode, loss
        sum_stacks mean((out - gaussian(pts))^2) with the target generated on the fly from `pts`
        ([B][16][2] heat-map coordinates), backward into flat_grads.  Returns (loss 0-d GPU tensor, outputs).
        The loss tensor is a VIEW of one slot of a ring of 16 device floats the engine writes (pa_hg_set_loss_total: no framework
        reduction between backward and optimizer): it holds this step's loss until the 16th later call of this method overwrites
        it -- consume it (meters, .item(), .clone()) before that; the shipped loops read it in the same iteration.
        dropout_masks ([B][1][4][4]): the occlusion masks of the reference's dropout branch, in every stack.
        after_forward: a callable run between the two passes; its accuracy() / pckh_origin_res() calls (stack-hg.py:176-178 read
        nothing but the forward pass's heat maps) go to the engine's meter stream and run BESIDE the backward pass instead of between
        two steps (pa_net_meters_async); their results are ordered behind this call like everything else.  The value it returns is
        kept in self.after_forward_result.  (Not with use_graph: there the two passes are one launch.)"""
        B = x.shape[0] if x is not None else img4.shape[0]
        h = self._net(B)
        self._last_B = B
        p = pts.to(torch.float64).contiguous()
        losses = torch.empty(self.num_stacks, dtype=torch.float32, device=self.flat_params.device)      # the engine copies every entry
        if self.use_graph and x is None and dropout_masks is None:
            check(lib().pa_hg_train_step(h, ptr(img4), ptr(p), 1 if self.training else 0, 1, ptr(losses)), 'pa_hg_train_step')
            if self.training:
                self._nbt += 1
            return losses.sum(), (self.heatmaps(B) if want_outputs else None)
        keep = self._set_masks(h, dropout_masks)
        # the sum over the stacks comes from the engine (pa_hg_set_loss_total): one of 16 rotating device floats, valid until 16 calls later
        if getattr(self, '_loss_ring', None) is None:
            self._loss_ring, self._loss_slot = torch.zeros(16, dtype=torch.float32, device=self.flat_params.device), 0
        self._loss_slot = (self._loss_slot + 1) % 16
        total = self._loss_ring[self._loss_slot:self._loss_slot + 1]
        check(lib().pa_hg_set_loss_total(h, ptr(total)), 'pa_hg_set_loss_total')
        try:
            check(lib().pa_hg_forward(h, ptr(x.contiguous().float()) if x is not None else None, ptr(img4), ptr(p),
                                      1 if self.training else 0, ptr(losses)), 'pa_hg_forward')
            if self.training:
                self._nbt += 1
            if after_forward is not None:
                self._meter_keep = []                           # buffers the meter stream uses: alive until the backward pass has joined it
                check(lib().pa_net_meters_async(h, 1), 'pa_net_meters_async')
                try:
                    self.after_forward_result = after_forward()
                finally:
                    check(lib().pa_net_meters_async(h, 0), 'pa_net_meters_async')
            if self.on_stack_done is not None:                  # the backward pass in phases: a finished stack's gradients can travel
                for phase in range(self.num_stacks + 1):
                    check(lib().pa_hg_backward_phase(h, phase), 'pa_hg_backward_phase')
                    if phase < self.num_stacks:
                        self.on_stack_done(self.num_stacks - 1 - phase)
            else:
                check(lib().pa_hg_backward(h), 'pa_hg_backward')
            self._meter_keep = None                             # (the backward pass has enqueued the join: later users of these blocks are ordered behind the meters)
            outs = self.heatmaps(B) if want_outputs else None
        finally:
            lib().pa_hg_set_loss_total(h, None)
            if keep is not None:
                self._set_masks(h, None)
            if getattr(self, '_meter_keep', None) is not None:
                # after_forward() or the backward call raised: the join was never enqueued.  Wait for the meter stream before its
                # buffers go back to the allocator, and stop collecting (later accuracy() calls would grow the list for ever).
                torch.cuda.synchronize()
                self._meter_keep = None
        return total.view(()), outs

    def accuracy(self, idxs, stack=-1):
        """Evaluation.accuracy (pylib/Evaluation.py:54-75) of the last forward's heat maps against the
        Gaussian target of the joints it was given, entirely on the device."""
        B = self._last_B
        h = self._net(B)
        stack = stack % self.num_stacks
        Hh = self.res // 4
        dev = self.flat_params.device
        scratch = torch.empty(B * 16 * Hh * Hh + 4 * B * 16 + B + 16, dtype=torch.float32, device=dev)
        # cached: a host->device copy here would make the host wait for the whole step (no run-ahead)
        key = tuple(int(i) for i in idxs)
        cache = self.__dict__.setdefault('_idx_cache', {})
        ix = cache.get(key)
        if ix is None:
            ix = cache[key] = torch.as_tensor(key, dtype=torch.int32, device=dev)
        acc = torch.empty(len(idxs) + 1, dtype=torch.float32, device=dev)          # pck_kernel writes every entry
        check(lib().pa_hg_accuracy(h, stack, ptr(ix), len(idxs), ptr(acc), ptr(scratch)), 'pa_hg_accuracy')
        if getattr(self, '_meter_keep', None) is not None:
            self._meter_keep += [scratch, acc, ix]
        return acc


    def pckh_origin_res(self, center, scale, rot, grnd_pts, normalizers, stack=-1, per_person=False):
        """Evaluation.accuracy_origin_res (pylib/Evaluation.py:77-97) -- and per_person_pckh (:99-167) when
        per_person -- of the last forward's heat maps, on the device.  Returns (acc[15], person[B] or None)."""
        from ..pylib.Evaluation import PCKH_JOINTS
        B = self._last_B
        h = self._net(B)
        stack = stack % self.num_stacks
        Hh = self.res // 4
        dev = self.flat_params.device
        if not hasattr(self, '_pckh_idx'):
            self._pckh_idx = torch.as_tensor(PCKH_JOINTS, dtype=torch.int32, device=dev)
        scratch = torch.empty(6 * B * 16 + (B * 16 * Hh * Hh if per_person else 0) + 16, dtype=torch.float32, device=dev)
        acc = torch.empty(len(PCKH_JOINTS) + 1, dtype=torch.float32, device=dev)
        person = torch.empty(B, dtype=torch.float32, device=dev) if per_person else None
        c = center.float().contiguous(); s = scale.float().contiguous(); r = rot.float().contiguous()
        g = grnd_pts.float().contiguous(); nm = normalizers.float().contiguous()
        check(lib().pa_hg_pckh(h, stack, ptr(c), ptr(s), ptr(r), ptr(g), ptr(nm), ptr(self._pckh_idx), len(PCKH_JOINTS),
                               ptr(acc), ptr(person), ptr(scratch)), 'pa_hg_pckh')
        if getattr(self, '_meter_keep', None) is not None:
            self._meter_keep += [scratch, acc, person, c, s, r, g, nm]
        return acc, person


def sample_mask(pred_masks, dropout_num=2, seed=0, step=0, uniforms=None):
    """_Hourglass._sample_mask (models/asn_stacked_hg.py:102-136) on the device: softmax over the 16 cells of each sample's
    [B][1][4][4] mask logits, `dropout_num` distinct cells drawn with those probabilities (the law of
    np.random.choice(replace=False); the engine's own counter-based stream, or `uniforms` [B][dropout_num] float64).
    Returns (masks [B][1][4][4] fp32 with zeros at the drawn cells, indexes [B][dropout_num] int64)."""
    assert pred_masks.dim() == 4 and pred_masks.shape[1] == 1 and pred_masks.shape[2] == pred_masks.shape[3]
    B, cells = pred_masks.shape[0], pred_masks.shape[2] * pred_masks.shape[3]
    dev = pred_masks.device
    lg = pred_masks.reshape(B, cells).float().contiguous()
    masks = torch.empty((B, cells), dtype=torch.float32, device=dev)
    idx = torch.empty((B, dropout_num), dtype=torch.int32, device=dev)
    u = uniforms.to(dev, torch.float64).contiguous() if uniforms is not None else None
    check(lib().pa_sample_dropout_masks(ptr(lg), B, cells, dropout_num, int(seed), int(step), ptr(u), None, ptr(masks), ptr(idx),
                                        stream()), 'pa_sample_dropout_masks')
    return masks.view(pred_masks.shape), idx.long()


def dropout(x, masks):
    """_Hourglass._dropout (models/asn_stacked_hg.py:79-100) on an NCHW fp32 tensor (operator-level entry; the networks apply
    it inside pa_hg_forward): the [B][1][4][4] cell mask, nearest-upsampled to the map, times every channel."""
    B, Cc, H, W = x.shape
    pad = (-Cc) % 8
    xh = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, pad)).to(_lib.act_dtype()).contiguous()
    out = torch.empty_like(xh)
    m = masks.to(x.device, torch.float32).reshape(B, 16).contiguous()
    check(lib().pa_cell_mask(ptr(xh), ptr(m), ptr(out), B, H, W, Cc + pad, stream()), 'pa_cell_mask')
    return out[..., :Cc].permute(0, 3, 1, 2).float()


def create_hg(num_stacks, num_modules, num_classes, chan, res=256, default_batch=24):
    """models/asn_stacked_hg.py:344-347."""
    return HourglassNet(num_modules=num_modules, num_stacks=num_stacks, chan=chan, num_classes=num_classes,
                        res=res, default_batch=default_batch)


def create_asn(chan_in, chan_out, scale_num=None, rotation_num=None, is_aug=False, is_dropout=False, res=256,
               default_batch=24):
    """models/asn_stacked_hg.py:441-444: the scale/rotation agent (is_aug) or the occlusion agent (is_dropout)."""
    from .asn import ASN
    assert is_aug != is_dropout                              # :351
    if chan_in != chan_out:
        raise ValueError('the reference always builds the agent with chan_in == chan_out')
    return ASN(chan_out, scale_num, rotation_num, res=res, default_batch=default_batch, is_dropout=is_dropout)
