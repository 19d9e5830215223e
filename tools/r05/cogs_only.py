#!/usr/bin/env python3
"""The co-gs leg of bench.py's training record alone (BASELINE config 5: 3 M Gaussians, 4K, depth on the training path):
python tools/r05/cogs_only.py [iters] -- one line with the rate and the per-phase medians (A/B of library builds via
GSR_LIBRARY)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from harness.train import train  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
torch.cuda.set_device(0)
res = train(bench.cogs_3m_4k(iters), torch.device("cuda", 0), 0, 1)
print("cogs", os.environ.get("GSR_LIBRARY", "default").split("_")[-1], "it/s", round(res["iters"] / res["seconds"], 1),
      res.get("phase_ms_median"), "N end", res.get("num_gaussians_end"))
