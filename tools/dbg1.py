import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, torch.nn.functional as F
from tests.test_gpu_conv import run_conv, rel_rms
torch.manual_seed(0)
# identity weights 64->64, 1x1: y should equal bf16(x)
B,C,H,W=1,64,8,8
x=torch.randn(B,C,H,W)
w=torch.eye(64).view(64,64,1,1).contiguous()
y,_=run_conv(0,x,None,w,None,B,C,C,H,W,1)
ref=x.bfloat16().float()
print('identity rel', rel_rms(y,ref))
if rel_rms(y,ref)>1e-3:
    # find permutation structure: for pixel 0, which input channel does each output channel show?
    xm=x.bfloat16().float().permute(0,2,3,1).reshape(-1,64)   # [m][c]
    ym=y.permute(0,2,3,1).reshape(-1,64)
    for m in range(3):
        print('m',m,'y[:8]',ym[m,:8].tolist()); print('    x[:8]',xm[m,:8].tolist())
    # match y[m][n] to x[m'][c']
    flat=xm.flatten()
    for n in range(0,8):
        v=ym[0,n]; idx=(flat==v).nonzero().flatten().tolist()
        print('y[0][%d]=%g found at'%(n,float(v)), [(i//64,i%64) for i in idx[:4]])
    for m in range(1,4):
        v=ym[m,0]; idx=(flat==v).nonzero().flatten().tolist()
        print('y[%d][0]=%g found at'%(m,float(v)), [(i//64,i%64) for i in idx[:4]])
