"""cProfile of the trainer harness' host side (who is waiting for whom?).  python tools/exp/prof_train.py"""
import cProfile, pstats, sys, os, io
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT, os.path.join(ROOT,"gaussian-splatting-toolkit_amd")]
import torch
from harness.train import TrainConfig, train
cfg=TrainConfig(num_gaussians=1_000_000,width=1920,height=1080,num_views=16,iters=100,sh_degree_interval=25)
dev=torch.device("cuda:0")
train(TrainConfig(num_gaussians=100_000,width=640,height=360,num_views=4,iters=5),dev)
pr=cProfile.Profile(); pr.enable()
res=train(cfg,dev)
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
print(res["iters_per_s"])
