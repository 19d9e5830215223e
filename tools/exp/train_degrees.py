import os, sys
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"gaussian-splatting-toolkit_amd")]
import torch
from harness.train import TrainConfig, train
for deg in (0, 1, 2):
    cfg = TrainConfig(num_gaussians=60_000, init_gaussians=20_000, width=320, height=208, num_views=8, iters=1300,
                      sh_degree=deg, sh_degree_interval=300, densify=True, densify_grad_thresh=0.001) if hasattr(TrainConfig, "densify_grad_thresh") else None
    if cfg is None:
        cfg = TrainConfig(num_gaussians=60_000, init_gaussians=20_000, width=320, height=208, num_views=8, iters=1300,
                          sh_degree=deg, sh_degree_interval=300, densify=True)
    r = train(cfg, torch.device("cuda:0"))
    print(deg, r["num_gaussians_start"], "->", r["num_gaussians_end"], round(r["psnr_start"],2), "->", round(r["psnr_end"],2), len(r["refinements"]))
