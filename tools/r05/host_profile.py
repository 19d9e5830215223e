#!/usr/bin/env python3
"""cProfile of config 3's first 2 000 iterations (all at 480 x 270: the host-bound phase): where the host's time goes.
python tools/r05/host_profile.py [iters] > profile.txt"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from harness.train import train  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.cuda.set_device(0)
cfg = bench.config3(iters)
cfg.eval_views = 0 if hasattr(cfg, "eval_views") else None
pr = cProfile.Profile()
pr.enable()
res = train(cfg, torch.device("cuda", 0), 0, 1)
pr.disable()
print("iters/s", round(res["iters"] / res["seconds"], 1), res.get("phase_ms_median"))
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35)
    print(s.getvalue()[:9000])
