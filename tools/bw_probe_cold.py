"""What plain streaming kernels reach on tensors of the step's sizes when their operands are NOT in the Infinity Cache: 640 MB of
other traffic between launches, one event pair per launch (the protocol of pa_conv2d_time's cold variants)."""
import torch
dev = 'cuda'
scratch = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
def cold(fn, n=20):
    tot = 0.0
    for i in range(n):
        scratch.fill_(i & 1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
def warm(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (25, 50, 100, 200):
    n = mb << 19                         # bf16 elements
    a = torch.randn(n, device=dev).to(torch.bfloat16); b = torch.randn(n, device=dev).to(torch.bfloat16); c = torch.empty_like(a)
    for name, fn, nbytes in (('copy a->c      ', lambda: c.copy_(a), 2 * mb), ('add a+b->c     ', lambda: torch.add(a, b, out=c), 3 * mb),
                             ('relu a->c      ', lambda: torch.relu(a, out=c) if False else torch.clamp_min(a, 0, out=c), 2 * mb),
                             ('sum a          ', lambda: a.float().sum() if False else torch.sum(a), mb), ('fill c         ', lambda: c.fill_(1.0), mb)):
        tc, tw = cold(fn), warm(fn)
        print('%4d MB tensors  %s cold %6.1f us = %5.2f TB/s   warm %6.1f us = %5.2f TB/s' % (mb, name, tc, nbytes * 1.048576 / tc, tw, nbytes * 1.048576 / tw), flush=True)
