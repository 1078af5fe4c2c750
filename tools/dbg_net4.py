import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, time
from oracle import model as om, step as ostep, pylib as opl
from tests import inputs, bf16_emul
from tests.test_gpu_net import rel_rms, cosine, t
from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
torch.set_num_threads(8)
for (stacks,B,res,chan,damp) in [(1,4,128,128,0.2),(2,4,128,128,0.2),(2,4,128,128,0.1),(2,2,256,256,0.2)]:
    ref = om.create_hg(stacks,1,16,chan); om.deterministic_fill_(ref, seed=7)
    with torch.no_grad():
        for n,p in ref.named_parameters():
            if n.endswith('bn3.weight'): p.mul_(damp)
    net = create_hg(stacks,1,16,chan,res=res,default_batch=B); net.load_state_dict(ref.state_dict())
    img = t(inputs.images(8, B, res)); pts = inputs.heat_pts(9, B, res=res // 4)
    heat = t(inputs.heatmaps_from_pts(pts, res=res // 4))
    ref.train(); net.train()
    outs_e = bf16_emul.emul_hourglass_net(ref, img)
    loss_e = opl.stack_mse(outs_e, heat); ref.zero_grad(); loss_e.backward()
    ge = {n: p.grad.clone() for n,p in ref.named_parameters()}
    out_o, loss_o = ostep.pose_loss_and_grads(ref, img, heat)
    loss, outs = net.loss_and_backward(img.cuda(), t(pts).cuda(), want_outputs=True)
    print('cfg',stacks,B,res,chan,damp,'loss hip %.6f emul %.6f oracle %.6f'%(float(loss),float(loss_e),float(loss_o)))
    print('  out err vs emul',[round(rel_rms(o.cpu(), r.detach()),4) for o,r in zip(outs,outs_e)], 'vs oracle',[round(rel_rms(o.cpu(), r.detach()),4) for o,r in zip(outs,out_o)])
    for tag,gref in (('emul',ge),('oracle',{n:p.grad for n,p in ref.named_parameters()})):
        rows=[]
        for name, g in net.named_grads():
            r = gref[name]
            if name.endswith('.bias') and float(r.abs().max()) < 1e-6*float(r.numel()**0.5): continue
            rows.append((rel_rms(g.cpu(), r), cosine(g.cpu(), r), name))
        rows.sort(reverse=True)
        allg = torch.cat([g.flatten().cpu() for _,g in net.named_grads()]); allr = torch.cat([gref[n].flatten() for n,_ in net.named_grads()])
        print('  vs %-6s median rel %.4f min cos %.5f | whole rel %.4f cos %.5f | worst %s'%(tag,np.median([r[0] for r in rows]), min(r[1] for r in rows), rel_rms(allg,allr), cosine(allg,allr), [(round(a,3),round(b,4),n) for a,b,n in rows[:4]]))
