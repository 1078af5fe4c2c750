"""A recording stand-in for libposeadv_hip.so, for CPU tests of the HOST control flow only (what is called, in which order,
with which arguments, on which rank).  It computes nothing of the product: every pa_* call is logged and returns 0; the few
calls the control flow depends on (state-dict table, gradient fill, RMSprop arithmetic on the flat arrays) are emulated on
CPU tensors.  Installed by `install()` into the already imported modules of the package -- test infrastructure, never
reachable from the product path."""
import sys

import torch

N_PARAMS, N_BUFFERS = 40, 8
TABLE = [('conv1.weight', (2, 3, 2, 2), 0, 24, 0), ('conv1.bias', (2,), 24, 2, 0), ('bn1.weight', (4,), 28, 4, 0),
         ('bn1.bias', (4,), 32, 4, 0), ('bn1.running_mean', (4,), 0, 4, 1), ('bn1.running_var', (4,), 4, 4, 1),
         ('bn1.num_batches_tracked', (), 0, 1, 2)]


class FakeLib(object):
    def __init__(self, rank):
        self.rank, self.log, self.nets = rank, [], {}

    def __getattr__(self, name):
        if not name.startswith('pa_'):
            raise AttributeError(name)

        def call(*args):
            self.log.append((name, args))
            return 0
        return call

    def names(self):
        return [n for n, _ in self.log]

    # ---- what the host layer needs answers from
    def pa_last_error(self):
        return b'stub'

    def pa_grad_scale(self):
        return 1.0

    def pa_dtype(self):
        return 0

    def pa_hg_create(self, *a):
        self.log.append(('pa_hg_create', a))
        self.nets[len(self.nets) + 1] = {}
        return len(self.nets)

    pa_asn_create = pa_hg_create

    def pa_net_num_tensors(self, h):
        return len(TABLE)

    def pa_net_tensor_info(self, h, i, name, cap, shape, nd, off, numel, kind):
        n, shp, o, ne, k = TABLE[i]
        name.value = n.encode()
        for j, v in enumerate(shp):
            shape[j] = v
        nd._obj.value, off._obj.value, numel._obj.value, kind._obj.value = len(shp), o, ne, k
        return 0

    def pa_net_param_floats(self, h):
        return N_PARAMS

    def pa_net_buffer_floats(self, h):
        return N_BUFFERS

    def pa_net_workspace_bytes(self, h):
        return 256

    def pa_crop_workspace_bytes(self, *a):
        return 256

    def pa_net_bind(self, h, params, grads, buffers, ws, stream):
        self.log.append(('pa_net_bind', (h, stream)))
        self.nets[h].update(params=params, grads=grads, buffers=buffers, stream=stream)
        return 0

    def pa_sample_aug(self, meta, si, ri, mode, seed, step, B, params, stream):
        self.log.append(('pa_sample_aug', (mode, seed, step, B)))
        g = torch.Generator().manual_seed(int(seed) * 1000003 + int(step))
        params.copy_(torch.rand(params.shape, generator=g, dtype=torch.float64))
        return 0

    def pa_hg_set_loss_total(self, h, total):
        self._loss_total = total
        return 0

    def pa_hg_forward(self, h, img, img4, pts, train, losses):
        self.log.append(('pa_hg_forward', (h, train)))
        if losses is not None:
            losses.fill_(0.5 + self.rank)
            if getattr(self, '_loss_total', None) is not None:
                self._loss_total.fill_(float(losses.sum()))
        return 0

    def pa_hg_backward(self, h):
        self.log.append(('pa_hg_backward', (h,)))
        g = torch.Generator().manual_seed(77 + self.rank + 10 * len([1 for n in self.names() if n == 'pa_hg_backward']))
        self.nets[h]['grads'].copy_(torch.randn(N_PARAMS, generator=g))          # every rank: its own shard's gradient
        return 0

    def pa_rmsprop_step(self, p, g, v, n, lr, alpha, eps, gscale, stream):
        self.log.append(('pa_rmsprop_step', (n, lr, alpha, eps, gscale, stream)))
        gg = g * gscale
        v.mul_(alpha).addcmul_(gg, gg, value=1 - alpha)
        p.addcdiv_(gg, v.sqrt().add_(eps), value=-lr)
        return 0

    def pa_rmsprop_step_state(self, p, g, v, n, lr, alpha, eps, gscale, state, stream):
        assert state.numel() == 2 and state.dtype == torch.int32          # the optimizer's own {flag, skipped} pair
        return self.pa_rmsprop_step(p, g, v, n, lr, alpha, eps, gscale, stream)


def install(rank):
    """Route the package's ctypes layer to a FakeLib on CPU tensors.  Returns the FakeLib."""
    import pose_adv_aug_amd  # noqa: F401
    from pose_adv_aug_amd import _lib
    import pose_adv_aug_amd.stack_hg, pose_adv_aug_amd.data, pose_adv_aug_amd.pylib, pose_adv_aug_amd.utils.optim  # noqa: F401,E401
    import pose_adv_aug_amd.models.asn_stacked_hg  # noqa: F401
    fake = FakeLib(rank)
    patch = {'lib': lambda: fake, 'ptr': lambda t: t, 'stream': lambda: 12345, 'require_gpu': lambda: None,
             'device': lambda: torch.device('cpu')}
    _lib._lib = fake
    for mod in list(sys.modules.values()):
        if getattr(mod, '__name__', '').startswith('pose_adv_aug_amd'):
            for k, v in patch.items():
                if hasattr(mod, k):
                    setattr(mod, k, v)
    return fake
