#!/usr/bin/env python3
"""Host time spent INSIDE every custom autograd node of a training-shaped step (forward and backward bodies), and in the
native calls they make, on a small tile grid where the step is host-bound.  python tools/exp/op_host_time.py [N] [W] [H]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch

from harness import scene as S
from harness.pipeline import CameraTensors
from harness.train import GaussianParams, blob_scene, orbit_cameras

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
H = int(sys.argv[3]) if len(sys.argv) > 3 else 270
dev = torch.device("cuda:0")
cam = CameraTensors.from_numpy(orbit_cameras(4, W, H, radius=5.0)[1], dev)
model = GaussianParams(blob_scene(N, seed=0, kind="objects", scale_lo=0.003, scale_hi=0.008, objects=(140, 0.18, 0.45),
                                  extent=2.5, tex_cell=0.03), dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
from gs_fused import l1_ssim_loss

target = torch.rand(H, W, 3, device=dev)
acc = collections.defaultdict(float)
cnt = collections.Counter()


def wrap(cls, name):
    for which in ("forward", "backward"):
        fn = getattr(cls, which)

        def timed(*a, _fn=fn, _key=f"{name}.{which}", **k):
            t = time.perf_counter()
            try:
                return _fn(*a, **k)
            finally:
                acc[_key] += time.perf_counter() - t
                cnt[_key] += 1

        setattr(cls, which, staticmethod(timed))


import gs_fused.activations as GA
import gs_fused.loss as GL
import gs_fused.sh as GS
import rasterizer.cuda as RC
import importlib

RP = importlib.import_module("rasterizer.project_gaussians")
RP = sys.modules["rasterizer.project_gaussians"]
RR = sys.modules.get("rasterizer.rasterize") or importlib.import_module("rasterizer.rasterize")

wrap(GA._Activate, "activate")
wrap(RP._ProjectGaussians, "project")
wrap(RR._RasterizeGaussians, "rasterize")
for nm in dir(GS):
    c = getattr(GS, nm)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, "sh_split")
for nm in dir(GL):
    c = getattr(GL, nm)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, "loss")
orig_call = RC._call


def timed_call(fn_name, *a):
    t = time.perf_counter()
    try:
        return orig_call(fn_name, *a)
    finally:
        acc["native:" + fn_name] += time.perf_counter() - t
        cnt["native:" + fn_name] += 1


RC._call = timed_call
for m in (GA, GL, GS, RR):
    if hasattr(m, "_call"):
        m._call = timed_call


def step():
    for p in model.param_list():
        p.grad = None
    out = model.render(cam, bg, 3, retain_xys_grad=True, clamp_rgb=False)
    loss = l1_ssim_loss(out["rgb"], target, 0.2, clamp_pred=True)
    loss.backward()


for _ in range(30):
    step()
torch.cuda.synchronize()
acc.clear(), cnt.clear()
n = 300
t0 = time.perf_counter()
tf = 0.0
for _ in range(n):
    step()
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n * 1e6
print(f"{N} Gaussians {W}x{H}: {total:.1f} us per step (host-bound if the sum below is close to it)")
for k in sorted(acc, key=lambda k: -acc[k]):
    print(f"  {k:42s} {acc[k] / n * 1e6:8.1f} us per step  ({cnt[k] / n:.1f} calls)")
print("  sum of the node bodies: %.1f us" % (sum(v for k, v in acc.items() if not k.startswith("native:")) / n * 1e6))
