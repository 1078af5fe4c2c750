"""A/B of 1x1 kernel variants selected by a tuning-build environment switch read once per process:
    python tools/bench_conv1_ab.py PA_CONV1_OLD 0 1
Shapes: the bottleneck layers of the 2-stack net at 64x64 and 32x32, batch 24; warm and cold (640 MB of other traffic between launches)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, '.')
    import ctypes as C
    import torch
    from pose_adv_aug_amd._lib import lib, check, ptr, stream
    L = lib()
    L.pa_conv2d_time.restype = C.c_int
    L.pa_conv2d_time.argtypes = [C.c_int] * 9 + [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
    ws = torch.zeros(3 << 30, dtype=torch.uint8, device='cuda')
    for (Cin, Cout, H) in ((256, 128, 64), (128, 256, 64), (256, 128, 32), (128, 256, 32)):
        for mode in (0, 1):
            for v in (3, 7, 19, 23):
                ms = C.c_float()
                check(L.pa_conv2d_time(mode, v, 24, Cin, Cout, H, H, 1, 30, ptr(ws), C.byref(ms), stream()))
                print('%s %3d->%3d %2dx%2d var %2d %s  %7.1f us' % ('fwd  ' if mode == 0 else 'dgrad', Cin, Cout, H, H, v & 15, 'cold' if v & 16 else 'warm', ms.value * 1e3), flush=True)
else:
    var = sys.argv[1]
    for m in sys.argv[2:]:
        print('--- %s=%s' % (var, m), flush=True)
        subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, **{var: m}), check=True)
