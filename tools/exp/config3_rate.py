#!/usr/bin/env python3
"""BASELINE config 3 once (bench.config3: every reference default), nothing else: iterations/s and the phase medians
per resolution stage.  Environment knobs of the rasterizer apply -- for A/Bs of the small- and mid-grid rules.
    python tools/exp/config3_rate.py [iters] [export.ply]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from harness.train import train  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
cfg = bench.config3(iters)
if len(sys.argv) > 2:
    cfg.export_ply = sys.argv[2]
torch.cuda.set_device(0)
res = train(cfg, torch.device("cuda", 0), 0, 1)
print(json.dumps({"iters_per_s": round(res["iters_per_s"], 1), "seconds": round(res["seconds"], 2),
                  "gaussians_end": res["num_gaussians_end"], "psnr_end": round(res["psnr_end"], 2),
                  "by_resolution": res["phase_ms_median_by_resolution"]}))
