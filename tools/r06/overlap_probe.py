#!/usr/bin/env python3
"""How much would overlapping list construction (latency-bound, ~15 % of HBM peak) with compositing (VALU-bound) buy?
The upper bound, measured: the bench's 1 M / 1080p workload; the tile lists of a view built on a side stream WHILE the
compositing forward / backward of (the same lists) runs on the main stream, against the two run one after the other.
    python tools/r06/overlap_probe.py"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rasterizer.cuda as C  # noqa: E402
from harness import scene as S  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
W, H, n = 1920, 1080, 1_000_000
cam = S.make_camera(W, H)
sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=0.0025, scale_hi=0.025)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
    n, cu(sc["means3d"]), cu(sc["scales"]), 1.0, cu(sc["quats"]), cu(cam.viewmat[:3]), cu(cam.projmat),
    cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16, 0.01)
g = torch.Generator(device=dev).manual_seed(1)
colors = torch.rand(n, 3, device=dev, generator=g)
opac = cu(sc["opacities"]).reshape(n, 1)
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
v_img = torch.randn(H, W, 3, device=dev, generator=g)
v_alpha = torch.randn(H, W, device=dev, generator=g)
tb = ((W + 15) // 16, (H + 15) // 16, 1)
count = torch.zeros(1, dtype=torch.int32, device=dev)
cap = 6 << 20


def lists():
    return C.rasterize_gaussians_forward(xys, depths, radii, conics, None, opac, None, H, W, cap, count, composite=False, checked=True)


ids, bins = lists()
torch.cuda.synchronize()
assert int(count.item()) <= cap, int(count.item())


def fwd():
    return C.rasterize_forward_ex(tb, (16, 16, 1), (W, H, 1), ids, bins, xys, conics, colors, opac, bg, want_alpha=True)


img, Ts, idx, alpha = fwd()


def bwd():
    return C.rasterize_backward(H, W, 16, ids, bins, xys, conics, colors, opac, bg, Ts, idx, v_img, v_alpha)


side = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)


def timed(fn_main, fn_side, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(reps + 5):
        if k == 5:
            torch.cuda.synchronize()
            e0.record()
        if fn_side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                keep = fn_side()
        if fn_main is not None:
            fn_main()
        if fn_side is not None:
            main.wait_stream(side)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {"lists_alone": timed(lists, None), "fwd_alone": timed(fwd, None), "bwd_alone": timed(bwd, None),
       "lists_on_side_alone": timed(None, lists),
       "fwd_with_lists_on_side": timed(fwd, lists), "bwd_with_lists_on_side": timed(bwd, lists)}
for k, v in res.items():
    print(f"{k:28s} {v:8.1f} us")
for a in ("fwd", "bwd"):
    serial = res["lists_alone"] + res[a + "_alone"]
    both = res[a + "_with_lists_on_side"]
    print(f"{a}: serial {serial:.1f} us, concurrent {both:.1f} us -> hidden {serial - both:.1f} us of the lists' {res['lists_alone']:.1f} "
          f"(ideal: {serial - max(res['lists_alone'], res[a + '_alone']):.1f})")
