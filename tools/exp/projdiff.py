import sys
sys.path[:0]=['/root/repo','/root/repo/gaussian-splatting-toolkit_amd','/root/repo/tests']
import numpy as np, torch
import test_gpu_kernels as T
for (n,W,H,bw,ck) in T.CASES+[(1_000_000,1920,1080,16,{})]:
    cam, sc = T.make(n, W, H, cam_kw=ck, scale_lo=0.01, scale_hi=0.2)
    ref = T.project_cpu(cam, sc, bw)
    out = [T.npy(t) for t in T.project_gpu(cam, sc, bw)]
    names = ["cov3d", "xys", "depths", "radii", "conics", "compensation", "num_tiles_hit"]
    print(n, {k: (int((o!=r).sum()), float(np.abs(o.astype(np.float64)-r).max())) for k,o,r in zip(names,out,ref)})
