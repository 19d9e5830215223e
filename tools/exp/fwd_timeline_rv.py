#!/usr/bin/env python3
"""render_view under no_grad with a turning camera (tools/render_bench.py's loop), with the time spent waiting for the list
count measured: python tools/exp/fwd_timeline_rv.py <repo root to import from>"""
import os
import sys
import time

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import torch

from harness import scene as S
from harness.pipeline import CameraTensors, render_view
from rasterizer import rasterize as R

dev = torch.device("cuda:0")
cams = [S.make_camera(1920, 1080, yaw=0.01 * k) for k in range(8)]
sc = S.make_scene(1_000_000, cams[0], sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
p = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
camt = [CameraTensors.from_numpy(c, dev) for c in cams]
bg = torch.tensor(S.BACKGROUND, device=dev)
wait = {"t": 0.0, "n": 0}
orig = R._PendingCount.resolve


def timed(self):
    t = time.perf_counter()
    v = orig(self)
    wait["t"] += time.perf_counter() - t
    wait["n"] += 1
    return v


R._PendingCount.resolve = timed
keep = len(sys.argv) > 2 and sys.argv[2] == "keep"


def frame(k):
    with torch.no_grad():
        return render_view(p["means3d"], p["scales"], p["quats"], p["opacities"], p["sh_coeffs"], camt[k % 8], bg, 3)


for k in range(30):
    out = frame(k)
torch.cuda.synchronize()
wait.update(t=0.0, n=0)
n = 300
t0 = time.perf_counter()
for k in range(n):
    if keep:
        out = frame(k)
    else:
        frame(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e6
print(ROOT, "keep" if keep else "drop", "%.1f us per frame; in resolve %.1f us per frame (%d calls)" % (dt, wait["t"] / n * 1e6, wait["n"]))
