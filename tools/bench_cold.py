"""Hot (operands resident in the Infinity Cache) vs cold (640 MB of other traffic between launches) timing of the conv kernels."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
L.pa_conv2d_time.restype = C.c_int
L.pa_conv2d_time.argtypes = [C.c_int]*9 + [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
ws = torch.zeros(4 << 30, dtype=torch.uint8, device='cuda')
def run(mode, variant, B, Cin, Cout, H, W, k, iters=20):
    ms = C.c_float()
    check(L.pa_conv2d_time(mode, variant, B, Cin, Cout, H, W, k, iters, ptr(ws), C.byref(ms), stream()))
    return ms.value * 1e3
for sh in [(24,256,128,64,64,1),(24,128,256,64,64,1),(24,128,128,64,64,3),(24,256,128,32,32,1),(24,128,128,32,32,3)]:
    for mode in (0, 1, 2):
        for v in ((0, 3, 7) if mode < 2 else (1, 9)):
            hot, cold = run(mode, v, *sh), run(mode, v | 16, *sh)
            print('mode %d var %2d  %3d->%3d k%d %3dx%3d   hot %7.1f us   cold %7.1f us' % (mode, v, sh[1], sh[2], sh[5], sh[3], sh[4], hot, cold))
    print()
