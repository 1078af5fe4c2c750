"""Cycle stamps of the one-shot 1x1 kernel (tuning build, PA_CONV1_DBG=64): workgroup (0, 0), thread 0 -- shader cycles at
entry / loads issued / constants (finalize prologue) / staged / barrier / K loop / epilogue."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import os
os.environ['PA_CONV1_DBG'] = '64'
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
ws = torch.zeros(2 << 30, dtype=torch.uint8, device='cuda')
names = ['entry', 'loads issued', 'constants', 'staged', 'barrier', 'K loop', 'epilogue']
for mode, mname in ((0, 'fwd'), (1, 'dgrad')):
    for cin, cout in ((256, 128), (128, 256)):
        for H in (32, 16, 8, 4):
            ms = C.c_float()
            check(L.pa_conv2d_time(mode, 3, 24, cin, cout, H, H, 1, 20, ptr(ws), C.byref(ms), stream()))
            torch.cuda.synchronize()
            clk = (C.c_ulonglong * 32)()
            assert L.pa_debug_conv1_clocks(clk) == 0
            t = [clk[2 * i] for i in range(7)]; w = [clk[2 * i + 1] for i in range(7)]
            wall_us = (w[6] - w[0]) / 100.0
            mhz = (t[6] - t[0]) / wall_us if wall_us > 0 else 0
            print('%-5s %3d->%3d %2dx%-2d launch %5.1f us | in-kernel %5.2f us at %4.0f MHz | cycles: ' % (mname, cin, cout, H, H, ms.value * 1e3, wall_us, mhz) +
                  '  '.join('%s %d' % (names[i], t[i] - t[i - 1]) for i in range(1, 7)))
            e = [clk[2 * i] for i in range(7, 12)]
            if mode == 0 and all(e):
                print('        epilogue since the end of the K loop: pass0 staged %d, pass0 done %d, pass1 staged %d, pass1 done %d, shuffles done %d, end %d' % tuple([x - t[5] for x in e] + [t[6] - t[5]]))
