"""Does the depth sort overlap the SH kernel + caller glue when it runs on a high-priority side stream?
python tools/exp/overlap_sort.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch
import rasterizer.cuda as C
from harness import scene as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1920, 1080
dev = torch.device("cuda:0")
cam = S.make_camera(W, H)
sc = S.make_scene(n, cam, sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means, scales, quats, opac, sh = (cu(sc[k]) for k in ("means3d", "scales", "quats", "opacities", "sh_coeffs"))
viewmat, projmat, campos = cu(cam.viewmat[:3]), cu(cam.projmat), cu(cam.campos)
tb = ((W + 15) // 16, (H + 15) // 16, 1)
side = torch.cuda.Stream(device=dev, priority=-1)
cap = None


def frame(overlap):
    global cap
    cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
        n, means, scales, 1.0, quats, viewmat, projmat, cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16, 0.01)
    if overlap:
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            order, _ = C.depth_order(depths, radii, None)
            done = torch.cuda.Event()
            done.record(side)
    dirs = means - campos
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    rgbs = C.compute_sh_forward(n, 3, 3, dirs, sh)
    rgbs = torch.clamp(rgbs + 0.5, min=0.0)
    if overlap:
        torch.cuda.current_stream().wait_event(done)
    else:
        order, _ = C.depth_order(depths, radii, None)
    _, recs = C.count_reach(xys, radii, conics, opac, tb, counts=False)
    ids, bins = C.bin_sorted(n, cap, order, None, xys, radii, tb, 16, recs, device_sized=True)
    return ids, bins, rgbs


# capacity from the full flow once
cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
    n, means, scales, 1.0, quats, viewmat, projmat, cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16, 0.01)
cnt, _ = C.count_reach(xys, radii, conics, opac, tb)
_, cum = C.depth_order(depths, radii, cnt)
cap = int(1.3 * int(cum[-1].item()))
ref = None
for mode in (False, True, False, True):
    for _ in range(10):
        out = frame(mode)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        out = frame(mode)
    b.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    same = torch.equal(ref[1], out[1]) and torch.equal(ref[0], out[0])
    print(f"overlap={mode}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per (project + SH + glue + lists)  lists equal: {same}")
