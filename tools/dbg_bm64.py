import sys; sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from tests.test_gpu_conv import run_conv
from tests import inputs
B, Cin, Cout, H, W, k = 24, 128, 128, 64, 64, 3
g = inputs.rng(100, B, Cin, Cout, H, k)
x = torch.from_numpy(g.standard_normal((B, Cin, H, W)).astype(np.float32))
w = torch.from_numpy((g.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
y, _ = run_conv(0, x, None, w, None, B, Cin, Cout, H, W, k)
ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), None, padding=1)
err = (y - ref).abs()
bad = err > 0.05
print('bad frac', bad.float().mean().item(), 'max', err.max().item())
idx = bad.nonzero()
print('n bad', idx.shape[0])
if idx.shape[0]:
    print('images', torch.unique(idx[:, 0])[:20].tolist())
    print('channels', torch.unique(idx[:, 1])[:40].tolist(), len(torch.unique(idx[:, 1])))
    print('rows', torch.unique(idx[:, 2])[:70].tolist())
    print('cols', torch.unique(idx[:, 3])[:70].tolist())
