#!/usr/bin/env python3
"""BASELINE config 1 (10 k Gaussians, SH degree 0, 256 x 256) forward + backward: the public ops (render_view, the
models' call sequence) against the one-op path (gs_fused.render_gaussians).  Host-bound either way."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gs_fused import ViewSpec, render_gaussians
from harness import scene as S
from harness.pipeline import CameraTensors, render_view

dev = "cuda:0"
W = H = 256
n = 10_000
cam = S.make_camera(W, H)
sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=0.005, scale_hi=0.05)
camt = CameraTensors.from_numpy(cam, dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = {k: t(v).requires_grad_(True) for k, v in sc.items()}
raw = {"means": t(sc["means3d"]), "scales": t(np.log(sc["scales"])), "quats": t(sc["quats"]),
       "opacities": t(np.log(sc["opacities"] / (1 - sc["opacities"]))), "features_dc": t(sc["sh_coeffs"][:, 0, :]),
       "features_rest": torch.zeros(n, 0, 3, device=dev)}
raw = {k: v.requires_grad_(True) for k, v in raw.items()}
v_img = torch.rand(H, W, 3, device=dev)
spec = ViewSpec(H, W, cam.fx, cam.fy, cam.cx, cam.cy, 0)

def public():
    for q in p.values():
        q.grad = None
    out = render_view(p["means3d"], p["scales"], p["quats"], p["opacities"], p["sh_coeffs"], camt, bg, 0, clamp_rgb=False)
    out["rgb"].backward(v_img)

def fused():
    for q in raw.values():
        q.grad = None
    out = render_gaussians(raw["means"], raw["scales"], raw["quats"], raw["opacities"], raw["features_dc"],
                           raw["features_rest"], camt.viewmat, camt.projmat, camt.campos, bg, spec, 1 << 20)
    out["rgb"].backward(v_img)

res = {}
for name, fn in (("public_ops", public), ("render_gaussians", fused)):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    res[name + "_ms"] = round((time.perf_counter() - t0) / 300 * 1e3, 4)
print(json.dumps(res))
