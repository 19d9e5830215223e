#!/usr/bin/env python3
"""Round 3, verdict item 5: the scan mapping of the compositing forward (lanes over splats, wave prefix
product) against the serial tile16 walk, kernel time on three scenes.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import rasterizer.cuda as C
from harness import scene as S

dev = "cuda:0"
res = {}
for name, (n, W, H, lo, hi, longtail) in {"default": (1_000_000, 1920, 1080, 0.0025, 0.025, False),
                                            "longtail": (1_000_000, 1920, 1080, 0.0025, 0.025, True),
                                            "config5": (3_000_000, 3840, 2160, 0.005, 0.05, False)}.items():
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=lo, scale_hi=hi, longtail=longtail)
    t = lambda a: torch.from_numpy(a).to(dev)
    cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
        n, t(sc["means3d"]), t(sc["scales"]), 1.0, t(sc["quats"]), t(cam.viewmat[:3].copy()), t(cam.projmat), cam.fx, cam.fy,
        cam.cx, cam.cy, H, W, 16, 0.01)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    opac = t(sc["opacities"])
    colors = torch.rand(n, 3, device=dev)
    bg = torch.tensor(S.BACKGROUND, device=dev)
    counts, recs = C.count_reach(xys, radii, conics, opac, tb)
    order, cum = C.depth_order(depths, radii, counts)
    I = int(cum[-1].item())
    ids, bins = C.bin_sorted(n, I, order, cum, xys, radii, tb, 16, recs)
    out = {}
    for key, fn in {"serial": lambda: C.rasterize_forward(tb, (16, 16, 1), (W, H, 1), ids, bins, xys, conics, colors, opac, bg),
                    "scan": lambda: C.rasterize_forward_scan(tb, (W, H, 1), ids, bins, xys, conics, colors, opac, bg)}.items():
        for _ in range(3):
            r = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        out[key + "_ms"] = round(e0.elapsed_time(e1) / 10, 4)
        out[key] = r
    d = (out["serial"][0] - out["scan"][0]).abs()
    res[name] = {"list_entries": I, "serial_ms": out["serial_ms"], "scan_ms": out["scan_ms"],
                 "max_abs_diff": float(d.max()), "frac_above_1e-4": float((d > 1e-4).float().mean())}
print(json.dumps(res))
