"""Micro-benchmark of single conv launches (run on the GPU box): time, TFLOP/s, algorithmic GB/s."""
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
shapes = [(24,256,128,64,64,1),(24,128,256,64,64,1),(24,128,128,64,64,3),(24,256,256,64,64,1)]
if len(sys.argv) > 1 and sys.argv[1] == 'all':
    shapes += [(24,128,128,32,32,3),(24,256,128,32,32,1),(24,128,128,16,16,3),(24,64,64,128,128,3),(24,64,128,128,128,1)]
for sh in shapes:
    for mode, variants in ((0, (0,1,2,3,4,7)), (1, (0,1,2,3,7)), (2, (0,1,8,9))):
        for v in variants:
            run(mode, v, *sh)
    print()
