"""Cycle stamps of the row-tile 1x1 kernel (tuning build): thread 0 of workgroup PA_STAMP_WG (default 0 = first round; 600 = a second-round
workgroup at 64x64) -- shader cycles at entry / first loads issued / staged / barrier / K loop (first channel block) / its epilogue / end."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import os
wg = int(os.environ.get('PA_STAMP_WG', '0'))
os.environ['PA_CONV1T_DBG'] = str(64 + 256 * wg)
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
L.pa_conv2d_time.restype = C.c_int
L.pa_conv2d_time.argtypes = [C.c_int] * 9 + [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
ws = torch.zeros(4 << 30, dtype=torch.uint8, device='cuda')
names = ['entry', 'loads issued', 'staged', 'barrier', 'K loop 0', 'epilogue 0', 'end']
for mode, mname in ((0, 'fwd'), (1, 'dgrad')):
    for cin, cout, Hs in ((256, 128, (64, 32)), (128, 256, (64, 32)), (64, 64, (128,)), (64, 128, (128,))):
        for H in Hs:
            for variant in ((0, 3, 7) if mode == 0 else (0, 3, 7)):
                for cold in (0, 16):
                    ms = C.c_float()
                    check(L.pa_conv2d_time(mode, variant | cold, 24, cin, cout, H, H, 1, 10, ptr(ws), C.byref(ms), stream()))
                    torch.cuda.synchronize()
                    clk = (C.c_ulonglong * 32)()
                    assert L.pa_debug_conv1t_clocks(clk) == 0
                    t = [clk[2 * i] for i in range(7)]; w = [clk[2 * i + 1] for i in range(7)]
                    wall_us = (w[6] - w[0]) / 100.0
                    print('%-5s %3d->%3d %2dx%-2d var %d %s launch %5.1f us | wg %d in-kernel %5.2f us | cycles: ' % (mname, cin, cout, H, H, variant, 'cold' if cold else 'hot ', ms.value * 1e3, wg, wall_us) +
                          '  '.join('%s %d' % (names[i], t[i] - t[i - 1]) for i in range(1, 7)))
