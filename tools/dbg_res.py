import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import model as om
from tests import inputs
from tests.test_gpu_net import rel_rms, cosine, t
from pose_adv_aug_amd._lib import lib, check, ptr, stream
torch.set_num_threads(8)
for C_,H,B in [(128,16,2),(256,8,3),(128,4,1),(256,64,2)]:
    blk = om.Residual(C_, C_); om.deterministic_fill_(blk, seed=3); blk.train()
    x = torch.relu(t(inputs.rng(4).standard_normal((B, C_, H, H)).astype(np.float32)))
    dy = t(inputs.rng(5).standard_normal((B, C_, H, H)).astype(np.float32))
    xr = x.bfloat16().float().requires_grad_(True)
    # oracle with intermediate capture
    y = blk(xr); y.backward(dy.bfloat16().float())
    params = torch.cat([p.detach().flatten() for p in blk.parameters()]).cuda()
    blk0 = om.Residual(C_, C_); om.deterministic_fill_(blk0, seed=3)
    bufs0 = torch.cat([b.flatten().float() for n, b in blk0.named_buffers() if 'num_batches' not in n]).cuda()
    ws = torch.zeros(lib().pa_residual_workspace_bytes(B, H, H, C_), dtype=torch.uint8, device='cuda')
    yd = torch.empty_like(x).cuda(); dxd = torch.empty_like(x).cuda(); gd = torch.zeros_like(params)
    xd, dyd = x.cuda(), dy.cuda()
    check(lib().pa_residual_fwd_bwd(ptr(xd), ptr(dyd), ptr(params), ptr(yd), ptr(dxd), ptr(gd), ptr(bufs0), B, C_, H, H, ptr(ws), stream()))
    print('case',C_,H,B,'fwd %.4f'%rel_rms(yd.cpu(), y.detach()), 'dx %.4f cos %.5f'%(rel_rms(dxd.cpu(), xr.grad), cosine(dxd.cpu(), xr.grad)))
    off=0
    for name,p in blk.named_parameters():
        n=p.numel(); got=gd[off:off+n].cpu().view_as(p); off+=n
        print('   %-14s rel %.4f cos %.5f |ref| %.3e |got| %.3e'%(name, rel_rms(got,p.grad), cosine(got,p.grad), float(p.grad.norm()), float(got.norm())))
