#!/usr/bin/env python3
"""Two-round lists on the config-5 scene: unfinished tiles, Gaussians kept and entries built against the prefix length."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import rasterizer.cuda as C
from harness import scene as S

dev = "cuda:0"
n, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 3840, 2160
cam = S.make_camera(W, H)
sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=0.005, scale_hi=0.05)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
    n, t(sc["means3d"]), t(sc["scales"]), 1.0, t(sc["quats"]), t(cam.viewmat[:3].copy()), t(cam.projmat), cam.fx, cam.fy,
    cam.cx, cam.cy, H, W, 16, 0.01)
tb = ((W + 15) // 16, (H + 15) // 16, 1)
nt = tb[0] * tb[1]
opac = t(sc["opacities"])
colors = torch.rand(n, 3, device=dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
order, _ = C.depth_order(depths, radii, None)
n_culled = int((radii <= 0).sum())
full = None
for frac in (0.05, 0.1, 0.2, 0.4):
    _, recs = C.count_reach(xys, radii, conics, opac, tb, counts=False, extra_rows=1)
    n1 = n_culled + int(frac * (n - n_culled))
    cap = 120_000_000
    both = torch.empty(2 * cap, dtype=torch.int32, device=dev)
    c1, c2 = (torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2))
    bins1 = C.tile_lists_subrange(order[:n1], cap, recs, tb, both[:cap], c1)
    flags = torch.zeros(nt, dtype=torch.int32, device=dev)
    img = torch.empty(H, W, 3, device=dev); Ts = torch.empty(H, W, device=dev); idx = torch.empty(H, W, dtype=torch.int32, device=dev)
    C.rasterize_forward_round(1, tb, (W, H, 1), both, bins1, 0, xys, conics, colors, None, opac, bg, 0.0, img, None, Ts, idx, flags)
    stats = torch.zeros(2, dtype=torch.int32, device=dev)
    order2 = C.saturation_filter(order[n1:], recs, n, flags, tb, stats)
    bins2 = C.tile_lists_subrange(order2, cap, recs, tb, both[cap:], c2)
    torch.cuda.synchronize()
    live_px = int((Ts > 0).sum())
    print(json.dumps({"prefix": frac, "entries_round1": int(c1[0]), "per_tile": round(int(c1[0]) / nt, 1),
                      "unfinished_tiles": int(stats[0]), "tiles": nt, "live_pixels": live_px,
                      "suffix_gaussians": n - n1, "kept": int(stats[1]), "entries_round2": int(c2[0])}))
