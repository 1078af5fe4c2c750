"""Phase timing of the 3x3 tile kernel at the low-resolution levels (24 x 16 x 16, 8 x 8, 4 x 4; 128 -> 128): tuning build, PA_CONV3_DBG bits
1 / 2 / 4 = no K loop / no epilogue / no staging (wrong results), forward with BatchNorm+ReLU on load + statistics (var 3)."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import os
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
ws = torch.zeros(2 << 30, dtype=torch.uint8, device='cuda')
def run(mode, variant, B, Cin, Cout, H, W, k, iters=50):
    ms = C.c_float()
    check(L.pa_conv2d_time(mode, variant, B, Cin, Cout, H, W, k, iters, ptr(ws), C.byref(ms), stream()))
    return ms.value * 1e3
tag = 'DBG=%s' % os.environ.get('PA_CONV3_DBG', '-')
for H in (32, 16, 8, 4):
    print('%-8s %2dx%-2d  fwd var3 hot %5.1f us cold %5.1f us | dgrad var3 hot %5.1f cold %5.1f' % (tag, H, H, run(0, 3, 24, 128, 128, H, H, 3), run(0, 19, 24, 128, 128, H, H, 3),
          run(1, 3, 24, 128, 128, H, H, 3), run(1, 19, 24, 128, 128, H, H, 3)))
