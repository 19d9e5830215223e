#!/usr/bin/env python3
"""cProfile of the host side of one view with the unchanged models' read-backs (the sequence of
tools/exp/sync_timeline.py): which Python functions the ~1 ms of host time per step is spent in.
    python tools/exp/host_profile.py [steps] [top]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import torch

from harness import scene as S
from harness.pipeline import CameraTensors
from rasterizer import project_gaussians, rasterize_gaussians, spherical_harmonics

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
dev = torch.device("cuda:0")
cam = S.make_camera(1920, 1080)
sc = S.make_scene(1_000_000, cam, sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
p = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
ct = CameraTensors.from_numpy(cam, dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
v_img, v_alpha = (torch.from_numpy(a).to(dev) for a in S.make_cotangents(cam))


def step():
    for t in p.values():
        t.grad = None
    xys, depths, radii, conics, comp, tiles, _c = project_gaussians(
        p["means3d"], p["scales"], 1, p["quats"], ct.viewmat[:3, :], ct.projmat, ct.fx, ct.fy, ct.cx, ct.cy, ct.height,
        ct.width, 16)
    assert not (radii.sum() == 0)
    d = p["means3d"].detach() - ct.campos
    d = d / d.norm(dim=-1, keepdim=True)
    rgbs = torch.clamp(spherical_harmonics(3, d, p["sh_coeffs"]) + 0.5, min=0.0)
    assert (tiles > 0).any()
    rgb, alpha = rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, p["opacities"], ct.height, ct.width, 16,
                                     background=bg, return_alpha=True)
    torch.autograd.backward([rgb, alpha], [v_img, v_alpha])


for _ in range(30):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(top)
st.sort_stats("cumulative").print_stats(top)
