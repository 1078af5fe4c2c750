"""Practical HBM rates on the box with plain torch kernels over 1.6 GB (cold by construction): copy, read-only, write-only."""
import torch, time
n = 1600 << 20
a = torch.empty(n, dtype=torch.uint8, device='cuda').view(torch.float32)
b = torch.empty_like(a)
def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
s = t(lambda: b.copy_(a)); print('copy   r+w %.2f TB/s' % (2 * n / s / 1e12))
s = t(lambda: a.sum()); print('sum    r   %.2f TB/s' % (n / s / 1e12))
s = t(lambda: b.fill_(1.0)); print('fill   w   %.2f TB/s' % (n / s / 1e12))
s = t(lambda: torch.add(a, b, out=b)); print('add  2r+w  %.2f TB/s' % (3 * n / s / 1e12))
h = a.view(torch.bfloat16)
s = t(lambda: h.float().sum()); print('bf16->f32 sum (r + w f32 + r) ...')
for mb in (25, 50, 100, 200, 400, 800):
    k = mb << 20
    x = a.view(torch.uint8)[:k].view(torch.float32); y = b.view(torch.uint8)[:k].view(torch.float32)
    s = t(lambda: y.copy_(x), 20); print('copy %4d MB (hot if it fits the 256 MB Infinity Cache): %.2f TB/s' % (mb, 2 * k / s / 1e12))
