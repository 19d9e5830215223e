#!/usr/bin/env python3
"""Host time per public op of a forward-only frame (no_grad): python tools/exp/fwd_timeline.py <repo root to import from>"""
import os
import sys
import time

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch

from harness import scene as S
from harness.pipeline import CameraTensors
from rasterizer import project_gaussians, rasterize_gaussians, spherical_harmonics

dev = torch.device("cuda:0")
cam = S.make_camera(1920, 1080)
sc = S.make_scene(1_000_000, cam, sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
p = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
ct = CameraTensors.from_numpy(cam, dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
names = ["project", "viewdirs_sh_clamp", "rasterize"]
acc = np.zeros(3)


def frame(rec):
    with torch.no_grad():
        ts = [time.perf_counter()]
        xys, depths, radii, conics, comp, tiles, _c = project_gaussians(
            p["means3d"], p["scales"], 1, p["quats"], ct.viewmat[:3, :], ct.projmat, ct.fx, ct.fy, ct.cx, ct.cy, ct.height,
            ct.width, 16)
        ts.append(time.perf_counter())
        d = p["means3d"] - ct.campos
        d = d / d.norm(dim=-1, keepdim=True)
        rgbs = torch.clamp(spherical_harmonics(3, d, p["sh_coeffs"]) + 0.5, min=0.0)
        ts.append(time.perf_counter())
        rgb, alpha = rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, p["opacities"], ct.height, ct.width, 16,
                                         background=bg, return_alpha=True)
        ts.append(time.perf_counter())
    if rec:
        acc[:] += np.diff(ts)


for _ in range(30):
    frame(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    frame(True)
torch.cuda.synchronize()
print(ROOT, "%.1f us per frame; host us: " % ((time.perf_counter() - t0) / n * 1e6) + ", ".join("%s %.1f" % (a, b / n * 1e6) for a, b in zip(names, acc)))
