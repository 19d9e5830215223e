import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT, os.path.join(ROOT,'gaussian-splatting-toolkit_amd')]
import numpy as np, torch
from harness import scene as S
from oracle import oracle as O
import rasterizer.cuda as C
DEV='cuda:0'
cu=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
W,H,n=1920,1080,200_000
cam=S.make_camera(W,H); sc=S.make_scene(n,cam,sh_degree=0,seed=42,scale_lo=0.005,scale_hi=0.05)
cov3d,xys,depths,radii,conics,comp,tiles=O.project_gaussians_forward(n,sc['means3d'],sc['scales'],1.0,sc['quats'],cam.viewmat[:3],cam.projmat,cam.fx,cam.fy,cam.cx,cam.cy,H,W,16,0.01)
tb=(120,68,1); I,cum=O.compute_cumulative_intersects(tiles)
_,_,ks,vs,bins=O.bin_and_sort_gaussians(n,I,xys,depths,radii,cum,tb,16)
rng=np.random.default_rng(1); colors=rng.uniform(0,1,(n,3)).astype(np.float32); bg=np.array(S.BACKGROUND,np.float32)
img,Ts,idx=O.rasterize_forward(tb,(16,16,1),(W,H,1),vs,bins,xys,conics,colors,sc['opacities'],bg)
v_img,v_alpha=S.make_cotangents(cam)
ref8=O.rasterize_backward(H,W,16,vs,bins,xys,conics,colors,sc['opacities'],bg,Ts,idx,v_img,v_alpha,with_abs_sums=True, ambig_eps=1e-5); ref=ref8[:4]; absr=ref8[4:8]; amb=ref8[8]; print('ambiguous gaussians', amb.mean())
for G in ('4','8'):
    os.environ['GSR_BWD_GROUP']=G
got=C.rasterize_backward(H,W,16,cu(vs),cu(bins),cu(xys),cu(conics),cu(colors),cu(sc['opacities']),cu(bg),cu(Ts),cu(idx),cu(v_img),cu(v_alpha))
for g,r,ab,nm in zip(got,ref,absr,['v_xy','v_conic','v_colors','v_opacity']):
    g=g.cpu().numpy(); err=np.abs(g-r); mx=np.abs(r).max(); print(nm,'max err/abs_sum all', (err/np.maximum(ab,1e-20)).max(), ' unambiguous only', (err/np.maximum(ab,1e-20))[~amb].max(), ' strict rel unamb', (np.maximum(err-2e-6*ab,0)/np.maximum(np.abs(r),1e-30))[~amb].max())
    floor=1e-3*mx; e=err/np.maximum(np.abs(r),floor)
    i=np.unravel_index(e.argmax(),e.shape)
    print(nm,'max|ref|',mx,'max rel(floor)',e.max(),'at',i,'ref',r[i],'got',g[i],'radius',radii[i[0]],'opac',sc['opacities'][i[0],0], ' L2 rel',np.linalg.norm(g-r)/np.linalg.norm(r), ' max abs err/max',err.max()/mx)
print('--- generic (nd) kernel vs oracle, and tile16 vs generic')
gotn=C.nd_rasterize_backward(H,W,16,cu(vs),cu(bins),cu(xys),cu(conics),cu(colors),cu(sc['opacities']),cu(bg),cu(Ts),cu(idx),cu(v_img),cu(v_alpha))
for g,gn,r,ab,nm in zip(got,gotn,ref,absr,['v_xy','v_conic','v_colors','v_opacity']):
    g=g.cpu().numpy(); gn=gn.cpu().numpy()
    e1=np.abs(gn-r)/np.maximum(ab,1e-20); e2=np.abs(g-gn)/np.maximum(ab,1e-20)
    i=np.unravel_index(e1.argmax(),e1.shape)
    print(nm,'generic vs oracle max err/abs',e1.max(),'at',i,' tile16 vs generic',e2.max())
# worst colour offender: per-Gaussian detail
gc=got[2].cpu().numpy(); e=np.abs(gc-ref[2])/np.maximum(absr[2],1e-20); i=np.unravel_index(e.argmax(),e.shape)[0]
print('worst gaussian',i,'xy',xys[i],'conic',conics[i],'radius',radii[i],'opac',sc['opacities'][i],'tiles',tiles[i],'depth',depths[i], 'ref',ref[2][i],'got',gc[i],'generic',gotn[2].cpu().numpy()[i])
