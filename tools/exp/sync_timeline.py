#!/usr/bin/env python3
"""Host-side timeline of one view with the unchanged models' read-backs (harness.pipeline.render_view's sequence,
restated with a clock between the steps): where the host spends its time between the head of a view and the end of
its backward.  python tools/exp/sync_timeline.py [off|on|camera] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch

from harness import scene as S
from harness.pipeline import CameraTensors
from rasterizer import project_gaussians, rasterize_gaussians, spherical_harmonics
from rasterizer import rasterize as R

mode = sys.argv[1] if len(sys.argv) > 1 else "on"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
cam = S.make_camera(1920, 1080)
sc = S.make_scene(1_000_000, cam, sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
p = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
ct = CameraTensors.from_numpy(cam, dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
v_img, v_alpha = (torch.from_numpy(a).to(dev) for a in S.make_cotangents(cam))
names = ["camera_items", "project", "sync1", "viewdirs_sh_clamp", "sync2", "rasterize", "backward"]
acc = np.zeros(len(names))


def step(record):
    for t in p.values():
        t.grad = None
    ts = [time.perf_counter()]
    if mode == "camera":
        s_ = ct.scalars
        _ = (s_[2].item(), s_[3].item(), float(s_[4] / (2 * s_[0])), float(s_[5] / (2 * s_[1])), s_[4].item(), s_[5].item(),
             s_[0].item(), s_[1].item())
    ts.append(time.perf_counter())
    xys, depths, radii, conics, comp, tiles, _c = project_gaussians(
        p["means3d"], p["scales"], 1, p["quats"], ct.viewmat[:3, :], ct.projmat, ct.fx, ct.fy, ct.cx, ct.cy, ct.height,
        ct.width, 16)
    ts.append(time.perf_counter())
    if mode != "off":
        assert not (radii.sum() == 0)
    ts.append(time.perf_counter())
    d = p["means3d"].detach() - ct.campos
    d = d / d.norm(dim=-1, keepdim=True)
    rgbs = torch.clamp(spherical_harmonics(3, d, p["sh_coeffs"]) + 0.5, min=0.0)
    ts.append(time.perf_counter())
    if mode != "off":
        assert (tiles > 0).any()
    ts.append(time.perf_counter())
    rgb, alpha = rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, p["opacities"], ct.height, ct.width, 16,
                                     background=bg, return_alpha=True)
    ts.append(time.perf_counter())
    torch.autograd.backward([rgb, alpha], [v_img, v_alpha])
    ts.append(time.perf_counter())
    if record:
        acc[:] += np.diff(ts)


for _ in range(20):
    step(False)
import gc

gc.collect()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / steps * 1e6
print(f"mode {mode} GSR_SPECULATE={os.environ.get('GSR_SPECULATE', 'auto')}: {total:.1f} us per step; host us per phase: "
      + ", ".join(f"{n} {a / steps * 1e6:.1f}" for n, a in zip(names, acc)) + f"; counters {R.counters}")
