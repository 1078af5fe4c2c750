"""Phase cycles of the 1x1 row-tile kernel (library built with PA_TUNING=1 and -DPA_C1_TIME in csrc/build.sh's flags): thread 0 of
every workgroup adds its shader-clock intervals to a device table: [between tiles, row request + transform into LDS (includes the
wait for the rows), wait + barrier, K loops, epilogues]; printed per workgroup-tile."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import torch
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
L.pa_conv2d_time.restype = C.c_int
L.pa_conv2d_time.argtypes = [C.c_int] * 9 + [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
L.pa_conv1_time_read.restype = C.c_int
L.pa_conv1_time_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ws = torch.zeros(3 << 30, dtype=torch.uint8, device='cuda')
names = ['between', 'rows->LDS', 'wait+bar', 'K loops', 'epilogues']
for (Cin, Cout, H) in ((256, 128, 64), (128, 256, 64), (256, 128, 32)):
    for mode in (0, 1):
        for v in (3, 19, 7 + 16):
            out = (C.c_ulonglong * 8)()
            check(L.pa_conv1_time_read(out, 1))
            ms = C.c_float()
            iters = 20
            check(L.pa_conv2d_time(mode, v, 24, Cin, Cout, H, H, 1, iters, ptr(ws), C.byref(ms), stream()))
            torch.cuda.synchronize()
            check(L.pa_conv1_time_read(out, 1))
            tiles = 24 * H * H // 64 * (iters + 3)
            wgs = out[7]
            print('%s %3d->%3d %2dx%2d var %2d %s %6.1f us | wgs/launch %5d | cycles per tile: ' % ('fwd  ' if mode == 0 else 'dgrad', Cin, Cout, H, H, v & 15, 'cold' if v & 16 else 'warm', ms.value * 1e3, wgs // (iters + 3))
                  + '  '.join('%s %6.0f' % (n, out[i] / tiles) for i, n in enumerate(names)), flush=True)
