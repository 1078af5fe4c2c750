import sys; sys.path.insert(0,'/root/repo')
import ctypes as C
import numpy as np, torch, time
from oracle import model as om, step as ostep, pylib as opl
from tests import inputs, bf16_emul
from tests.test_gpu_net import rel_rms, cosine, t, _hg_pair
from pose_adv_aug_amd._lib import lib, check, ptr
torch.set_num_threads(8)
def dbg(net, name, grad=0):
    h = net._net(net._last_B); shp=(C.c_int*4)()
    check(lib().pa_hg_debug_tensor(h, name.encode(), grad, None, shp))
    out = torch.empty(tuple(shp), device='cuda'); check(lib().pa_hg_debug_tensor(h, name.encode(), grad, ptr(out), shp)); return out.cpu()
stacks,B,res,chan = 1,2,128,128
ref, net = _hg_pair(stacks, chan, B, res, seed=7)
img = t(inputs.images(8, B, res)); pts = inputs.heat_pts(9, B, res=res // 4)
heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
ref.train(); net.train()
tap={}
outs_e = bf16_emul.emul_hourglass_net(ref, img, tap)
loss_e = opl.stack_mse(outs_e, heat); ref.zero_grad(); loss_e.backward()
loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
print('out err', rel_rms(outs[0].cpu(), outs_e[0].detach()))
for name,v in tap.items():
    d = dbg(net, name)
    line = '%-16s fwd rel %.4f' % (name, rel_rms(d, v.detach()))
    print(line)
