import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, time
from oracle import model as om, step as ostep, pylib as opl
from tests import inputs, bf16_emul
from tests.test_gpu_net import rel_rms, cosine, t, _hg_pair
torch.set_num_threads(8)
for (stacks,B,res,chan) in [(1,2,128,128),(2,2,128,128),(2,4,256,128)]:
    ref, net = _hg_pair(stacks, chan, B, res, seed=7)
    img = t(inputs.images(8, B, res)); pts = inputs.heat_pts(9, B, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    ref.train(); net.train()
    outs_e = bf16_emul.emul_hourglass_net(ref, img)
    loss_e = opl.stack_mse(outs_e, heat); ref.zero_grad(); loss_e.backward()
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    print('cfg',stacks,B,res,chan,'loss',float(loss),float(loss_e),'out err',[round(rel_rms(o.cpu(), r.detach()),4) for o,r in zip(outs,outs_e)])
    gref = dict(ref.named_parameters())
    rows=[]
    for name, g in net.named_grads():
        r = gref[name].grad
        if name.endswith('.bias') and float(r.abs().max()) < 1e-6: continue
        rows.append((rel_rms(g.cpu(), r), cosine(g.cpu(), r), name))
    order = {n:i for i,(n,_) in enumerate(net.named_grads())}
    print('  median rel %.4f  min cos %.5f'%(np.median([r[0] for r in rows]), min(r[1] for r in rows)))
    allg = torch.cat([g.flatten().cpu() for _,g in net.named_grads()]); allr = torch.cat([gref[n].grad.flatten() for n,_ in net.named_grads()])
    print('  whole-gradient rel %.4f cos %.5f'%(rel_rms(allg,allr), cosine(allg,allr)))
    if stacks==1 or True:
        for a,b,n in sorted(rows, key=lambda r: order[r[2]]):
            if n.endswith('weight') and ('conv' in n or 'adapter' in n or 'linear' in n) : print('   %-34s rel %.3f cos %.4f'%(n,a,b))
    break
