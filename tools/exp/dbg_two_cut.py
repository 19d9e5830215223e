import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import rasterizer.cuda as C
from harness import scene as S
dev = "cuda:0"
n, W, H = 200_000, 1920, 1080
cam = S.make_camera(W, H)
sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=0.01, scale_hi=0.06)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
    n, t(sc["means3d"]), t(sc["scales"]), 1.0, t(sc["quats"]), t(cam.viewmat[:3].copy()), t(cam.projmat), cam.fx, cam.fy,
    cam.cx, cam.cy, H, W, 16, 0.01)
tb = ((W + 15) // 16, (H + 15) // 16, 1); nt = tb[0] * tb[1]
opac = torch.full((n, 1), 0.1, device=dev)
colors = torch.rand(n, 3, device=dev); bg = torch.tensor(S.BACKGROUND, device=dev)
order, _ = C.depth_order(depths, radii, None)
nc = int((radii <= 0).sum())
for n1f, cap1, cap2 in ((0.1, 1 << 22, 1 << 24), (0.1, 1 << 20, 1 << 20), (0.3, 1 << 20, 1 << 21), (0.05, 1 << 20, 1 << 22)):
    n1 = ((nc + int(n1f * (n - nc))) + 255) & ~255
    _, recs = C.count_reach(xys, radii, conics, opac, tb, counts=False, extra_rows=1)
    both = torch.empty(cap1 + cap2, dtype=torch.int32, device=dev)
    c1, c2 = (torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2))
    bins1 = C.tile_lists_subrange(order[:n1], cap1, recs, tb, both[:cap1], c1)
    flags = torch.zeros(nt, dtype=torch.int32, device=dev)
    img = torch.empty(H, W, 3, device=dev); Ts = torch.empty(H, W, device=dev); idx = torch.empty(H, W, dtype=torch.int32, device=dev)
    al = torch.empty(H, W, device=dev)
    C.rasterize_forward_round(1, tb, (W, H, 1), both, bins1, 0, xys, conics, colors, None, opac, bg, 0.0, img, None, Ts, idx, flags, out_alpha=al)
    torch.cuda.synchronize(); print("round1 ok", int(c1[0]), cap1, int(bins1.max()), flush=True)
    stats = torch.zeros(2, dtype=torch.int32, device=dev)
    order2 = C.saturation_filter(order[n1:], recs, n, flags, tb, stats)
    bins2 = C.tile_lists_subrange(order2, cap2, recs, tb, both[cap1:], c2)
    torch.cuda.synchronize(); print("lists2 ok", int(c2[0]), cap2, int(bins2.max()), int(bins2.min()), stats.tolist(), flush=True)
    C.rasterize_forward_round(2, tb, (W, H, 1), both, bins2, cap1, xys, conics, colors, None, opac, bg, 0.0, img, None, Ts, idx, flags, out_alpha=al)
    torch.cuda.synchronize(); print("round2 ok", flush=True)
