#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "gaussian-splatting-toolkit_amd")]
import torch, bench
from harness.train import train
for name, kw in (("separate ops", {}), ("one op", {"fused_render": True}), ("hip graph", {"use_graph": True})):
    cfg = bench.config3(7000)
    cfg.phase_every = 0
    for k, v in kw.items():
        setattr(cfg, k, v)
    try:
        r = train(cfg, torch.device("cuda", 0))
        print(name, "it/s %.1f" % r["iters_per_s"], "N", r["num_gaussians_end"], "psnr %.2f" % r["psnr_end"], "overflow", r["list_overflow_views"], r["render"])
    except Exception as e:
        print(name, "FAILED", repr(e)[:300])
PY
