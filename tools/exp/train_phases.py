import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT, os.path.join(ROOT,'gaussian-splatting-toolkit_amd')]
import numpy as np, torch
from harness import scene as S
from harness.train import *
dev=torch.device('cuda',0)
cfg=TrainConfig(num_gaussians=1_000_000,width=1920,height=1080,num_views=8)
cams=[CameraTensors.from_numpy(c,dev) for c in orbit_cameras(8,1920,1080)]
bg=torch.tensor(S.BACKGROUND,device=dev)
model=GaussianParams(blob_scene(1_000_000,0),dev)
with torch.no_grad(): gt=[model.render(c,bg,3)['rgb'].clone() for c in cams]
optims={k:torch.optim.Adam([model.gauss[k]],lr=lr,eps=1e-15) for k,lr in LRS.items()}
fused=torch.optim.Adam([{'params':[model.gauss[k]],'lr':lr} for k,lr in LRS.items()],eps=1e-15,fused=True) if len(sys.argv)>1 else None
from gs_fused import l1_ssim_loss as fl
ev=lambda: torch.cuda.Event(enable_timing=True)
acc={}
for it in range(50):
    e=[ev() for _ in range(6)]
    for o in optims.values(): o.zero_grad(set_to_none=True)
    e[0].record()
    out=model.render(cams[it%8],bg,3,retain_xys_grad=True); rgb=out['rgb']
    e[1].record()
    loss = fl(rgb,gt[it%8],0.2) if fused else 0.8*(rgb-gt[it%8]).abs().mean()+0.2*(1-ssim(rgb,gt[it%8]))
    e[2].record()
    loss.backward()
    e[3].record()
    if fused: fused.step()
    else:
        for o in optims.values(): o.step()
    e[4].record()
    torch.cuda.synchronize()
    if it>=20:
        for nm,a,b in (('render_fwd',0,1),('loss_fwd',1,2),('backward_all',2,3),('adam',3,4)):
            acc.setdefault(nm,[]).append(e[a].elapsed_time(e[b]))
print({k:round(float(np.mean(v)),3) for k,v in acc.items()}, 'total', round(sum(float(np.mean(v)) for v in acc.values()),3),'ms')
