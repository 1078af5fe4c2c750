"""3x3 conv micro-benchmark at the low-resolution levels and the 32x32 level (the latency-bound variants of conv3x3_tile.hip).
Tuning build: PA_CONV3_SPS=0 selects one weight slice per K-loop step (round 1), default = a tap / half a tap per step."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
L.pa_conv2d_time.restype = C.c_int
L.pa_conv2d_time.argtypes = [C.c_int]*9 + [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
ws = torch.zeros(1 << 30, dtype=torch.uint8, device='cuda')
def run(mode, variant, B, Cin, Cout, H, W, k, iters=200):
    ms = C.c_float()
    check(L.pa_conv2d_time(mode, variant, B, Cin, Cout, H, W, k, iters, ptr(ws), C.byref(ms), stream()))
    print('mode %d var %2d  %3d->%3d k%d %3dx%3d  %8.1f us' % (mode, variant, Cin, Cout, k, H, W, ms.value*1e3))
for sh in [(24,128,128,32,32,3),(24,128,128,16,16,3),(24,128,128,8,8,3),(24,128,128,4,4,3)]:
    for mode, variants in ((0, (3,)), (1, (3,))):
        for v in variants:
            run(mode, v, *sh)
