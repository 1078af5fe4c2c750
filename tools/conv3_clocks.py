"""Cycle stamps of the 3x3 tile kernel (tuning build, PA_CONV3_DBG=64): workgroup (0, 0), wave 0 -- shader cycles (s_memtime) and the 100 MHz wall
clock at entry / after the weight-ring prologue / after its own halo staging / after the K loop / after the barrier / after the epilogue."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import os
os.environ['PA_CONV3_DBG'] = str(64 | int(os.environ.get('PA_CONV3_DBG', '0')))
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
ws = torch.zeros(2 << 30, dtype=torch.uint8, device='cuda')
names = ['entry', 'ring issued', 'staged', 'K loop', 'barrier', 'epilogue']
for mode, mname in ((0, 'fwd'), (1, 'dgrad')):
    for H in (64, 32, 16, 8, 4):
        ms = C.c_float()
        check(L.pa_conv2d_time(mode, 3, 24, 128, 128, H, H, 3, 20, ptr(ws), C.byref(ms), stream()))
        torch.cuda.synchronize()
        clk = (C.c_ulonglong * 64)()
        assert L.pa_debug_conv3_clocks(clk) == 0
        t = [clk[2 * i] for i in range(6)]; w = [clk[2 * i + 1] for i in range(6)]
        wall_us = (w[5] - w[0]) / 100.0
        mhz = (t[5] - t[0]) / wall_us if wall_us > 0 else 0
        print('%-5s %2dx%-2d launch %5.1f us | in-kernel %5.2f us at %4.0f MHz | cycles: ' % (mname, H, H, ms.value * 1e3, wall_us, mhz) +
              '  '.join('%s %d' % (names[i], t[i] - t[i - 1]) for i in range(1, 6)))
        if os.environ.get('PA_CLK_STEPS'):
            # per K-loop step of the plain loop: cycles from `staged` to the end of the step's DMA wait / to its barrier release
            print('        steps (body done = wait begins, barrier released) since staged: ' + ' '.join('(%d %d)' % (clk[17 + 2 * i] - t[2], clk[16 + 2 * i] - t[2]) for i in range(18) if clk[16 + 2 * i] > t[2]))
