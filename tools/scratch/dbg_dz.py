import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, ctypes as C
from pose_adv_aug_amd._lib import lib, check, ptr, stream
L = lib()
B, C_, H = 8, 128, 64
nws = L.pa_residual_workspace_bytes(B, H, H, C_)
SL = 256 << 20
big = torch.zeros(nws + 2 * SL, dtype=torch.uint8, device='cuda')
big[:SL] = 0x5A; big[SL + nws:] = 0x5A
ws = big[SL:SL + nws]
nparam = 0
x = torch.rand(B, C_, H, H, device='cuda'); dy = torch.randn(B, C_, H, H, device='cuda')
import oracle.model as om
blk = om.Residual(C_, C_); om.deterministic_fill_(blk, seed=3)
params = torch.cat([p.detach().flatten() for p in blk.parameters()]).cuda()
bufs0 = torch.cat([b.flatten().float() for n, b in blk.named_buffers() if 'num_batches' not in n]).cuda()
yd = torch.empty_like(x); dxd = torch.empty_like(x); gd = torch.zeros_like(params)
check(L.pa_residual_fwd_bwd(ptr(x), ptr(dy), ptr(params), ptr(yd), ptr(dxd), ptr(gd), ptr(bufs0), B, C_, H, H, C.c_void_p(ws.data_ptr()), stream()))
torch.cuda.synchronize()
lo = (big[:SL] != 0x5A).nonzero(); hi = (big[SL + nws:] != 0x5A).nonzero()
print('nws', nws, 'corrupt below', lo.numel(), 'above', hi.numel())
if hi.numel(): print('above offsets', int(hi.min()), int(hi.max()))
if lo.numel(): print('below offsets', int(lo.min()) - SL, int(lo.max()) - SL)
