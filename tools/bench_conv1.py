"""1x1 conv micro-benchmark (fwd mode 0, dgrad mode 1) over the shapes of the 2-stack net."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
L.pa_conv2d_time.restype = C.c_int
L.pa_conv2d_time.argtypes = [C.c_int]*9 + [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
ws = torch.zeros(3 << 30, dtype=torch.uint8, device='cuda')
def run(mode, variant, B, Cin, Cout, H, W, k, iters=30):
    ms = C.c_float()
    check(L.pa_conv2d_time(mode, variant, B, Cin, Cout, H, W, k, iters, ptr(ws), C.byref(ms), stream()))
    M = B*H*W
    fl = 2.0*M*Cin*Cout*k*k; by = 2.0*M*(Cin+Cout)
    print('mode %d var %2d  %3d->%3d k%d %3dx%3d  %8.1f us  %7.1f TF/s  %7.1f GB/s(alg)' % (mode, variant, Cin, Cout, k, H, W, ms.value*1e3, fl/ms.value/1e9, by/ms.value/1e6))
for sh in [(24,256,128,64,64,1),(24,128,256,64,64,1),(24,256,256,64,64,1),(24,256,128,32,32,1),(24,128,256,32,32,1),(24,64,128,128,128,1)]:
    for mode, variants in ((0, (0,1,3,7)), (1, (0,1,3,7))):
        for v in variants:
            run(mode, v, *sh)
    print()
