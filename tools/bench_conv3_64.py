"""3x3 convolution at the benchmark's 64 x 64 level (24 x 64 x 64, 128 -> 128): forward with BatchNorm+ReLU on load and statistics,
data gradient with the BatchNorm backward on load and the masked epilogue; hot (operands in the Infinity Cache) and cold (640 MB of other
traffic between launches).  Tuning builds: PA_CONV3_TRI=0/1 (three-tile workgroups), PA_CONV3_DBG bits 1 / 2 / 4 = no K loop / no
epilogue / no staging (wrong results: phase timing), 8 = per-workgroup tap rotation."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import os
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
ws = torch.zeros(4 << 30, dtype=torch.uint8, device='cuda')
def run(mode, variant, B, Cin, Cout, H, W, k, iters=30):
    ms = C.c_float()
    check(L.pa_conv2d_time(mode, variant, B, Cin, Cout, H, W, k, iters, ptr(ws), C.byref(ms), stream()))
    return ms.value * 1e3
tag = 'TRI=%s DBG=%s' % (os.environ.get('PA_CONV3_TRI', '-'), os.environ.get('PA_CONV3_DBG', '-'))
shapes = [(24, 128, 128, 64, 64, 3)] + ([(24, 128, 128, 32, 32, 3)] if '--all' in sys.argv else [])
for sh in shapes:
    for mode, v in ((0, 0), (0, 3), (1, 0), (1, 3)):
        hot, cold = run(mode, v, *sh), run(mode, v | 16, *sh)
        fl = 2.0 * sh[0] * sh[3] * sh[4] * sh[1] * sh[2] * 9
        print('%-14s mode %d var %d %3dx%-3d  hot %6.1f us (%5.0f TF/s)  cold %6.1f us (%5.0f TF/s)' % (tag, mode, v, sh[3], sh[4], hot, fl / hot / 1e6, cold, fl / cold / 1e6))
