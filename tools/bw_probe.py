"""HBM streaming calibration on the GPU box: torch copy / add kernels at the tensor sizes of the net."""
import torch, time
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mb in (25, 50, 100, 400, 1600):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device='cuda', dtype=torch.bfloat16); y = torch.empty_like(x); z = torch.randn_like(x)
    dt = t(lambda: y.copy_(x)); print('copy  %5d MB: %7.1f us  %6.2f TB/s (r+w)' % (mb, dt * 1e6, 2 * mb * 1.048576e6 / dt / 1e12))
    dt = t(lambda: torch.add(x, z, out=y)); print('add   %5d MB: %7.1f us  %6.2f TB/s (2r+w)' % (mb, dt * 1e6, 3 * mb * 1.048576e6 / dt / 1e12))
    dt = t(lambda: x.sum()); print('sum   %5d MB: %7.1f us  %6.2f TB/s (r)' % (mb, dt * 1e6, mb * 1.048576e6 / dt / 1e12))
