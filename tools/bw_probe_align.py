"""Does the relative placement of a kernel's operand tensors matter?  a + b -> c on three 24 MiB bf16 tensors (the size of a 24 x 64 x 64 x 128
activation) carved out of one buffer back to back (offsets multiples of 24 MiB, as the engine's arena lays activations out) or with a pad between them."""
import torch
dev = 'cuda'
N = 24 << 20                        # bytes per tensor
buf = torch.empty(4 * N + (64 << 20), dtype=torch.uint8, device=dev)
scratch = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
def view(off): return buf[off:off + N].view(torch.bfloat16)
def run(pad, cold, n=30):
    a, b, c = view(0), view(N + pad), view(2 * (N + pad))
    a.normal_(); b.normal_()
    fn = lambda: torch.add(a, b, out=c)
    fn(); torch.cuda.synchronize()
    if not cold:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tot = 0.0
    for i in range(n):
        scratch.fill_(i & 1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
for pad in (0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, (1 << 20) + 4096, (2 << 20), (3 << 20) + 8192):
    w, c = run(pad, False), run(pad, True)
    print('pad %8d B   warm %6.1f us = %5.2f TB/s   cold %6.1f us = %5.2f TB/s' % (pad, w, 3 * N / w / 1e6, c, 3 * N / c / 1e6), flush=True)
