"""Depth segments in the compositing backward (gsr_rasterize_backward_seg) against the single walk:
gradients (max error relative to max |ref| per tensor) and the time of the rasterizer's backward.
   python tools/exp/seg_ab.py [W H N] [reps]      BLOB=1: the trainer's object scene (deep centre tiles)"""
import gc, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch
import rasterizer.cuda as C
from rasterizer.rasterize import rasterize_gaussians
from harness import scene as S

W, H, n = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270, 300_000)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
if os.environ.get("BLOB"):
    from harness import train as T
    cam = T.orbit_cameras(16, W, H)[3]
    raw = T.blob_scene(n, seed=0, sh_degree=0)
    q = raw["quats"] / np.linalg.norm(raw["quats"], axis=-1, keepdims=True)
    sc = {"means3d": raw["means"], "scales": np.exp(raw["scales"]), "quats": q.astype(np.float32),
          "opacities": (1 / (1 + np.exp(-raw["opacities"]))).astype(np.float32)}
else:
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=float(os.environ.get("SCALE_LO", 0.005)),
                      scale_hi=float(os.environ.get("SCALE_HI", 0.03)))
cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
    n, cu(sc["means3d"]), cu(sc["scales"]), 1.0, cu(sc["quats"]), cu(cam.viewmat[:3]), cu(cam.projmat),
    cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16, 0.01)
g = torch.Generator(device="cuda").manual_seed(1)
colors = torch.rand(n, 3, device="cuda", generator=g)
opac = cu(sc["opacities"]).reshape(n, 1).clone()
if os.environ.get("OPAQUE"):
    opac = opac.clamp_min(float(os.environ["OPAQUE"]))
bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
v_img = torch.randn(H, W, 3, device="cuda", generator=g)
v_alpha = torch.randn(H, W, device="cuda", generator=g)
ntiles = ((W + 15) // 16) * ((H + 15) // 16)


def run(segs):
    C._segment_cache.clear()
    os.environ["GSR_DEPTH_SEGMENTS"] = str(segs)
    ins = [t.clone().requires_grad_(True) for t in (xys, conics, colors, opac)]
    img, alpha = rasterize_gaussians(ins[0], depths, radii, ins[1], tiles, ins[2], ins[3], H, W, 16, bg, return_alpha=True)
    torch.cuda.synchronize()
    gc.collect()  # (a full collection inside a 30-call loop is a 2 ms-per-call artefact)
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        f0.record()
        for _ in range(reps):
            rasterize_gaussians(xys, depths, radii, conics, tiles, colors, opac, H, W, 16, bg, return_alpha=True)
        f1.record()
    torch.cuda.synchronize()
    global t_fwd
    t_fwd = f0.elapsed_time(f1) / reps * 1e3
    loss = (img * v_img).sum() + (alpha * v_alpha).sum()
    grads = torch.autograd.grad(loss, ins, retain_graph=True)
    torch.cuda.synchronize()
    gc.collect()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.autograd.grad(loss, ins, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    return [t.detach() for t in grads], e0.elapsed_time(e1) / reps * 1e3, torch.cat([img.detach(), alpha.detach()[..., None]], -1)


ref, t1, img1 = run(1)
print(f"{W}x{H} ({ntiles} tiles), {n} Gaussians, visible {int((radii > 0).sum())}, intersections {int(tiles.sum())}")
print(f"segments 1: forward (lists + compositing) {t_fwd:8.1f} us  backward {t1:8.1f} us")
for segs in [int(x) for x in os.environ.get("SEGS", "2,4,8").split(",")]:
    got, t, img = run(segs)
    errs = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(got, ref)]
    l2 = [float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(got, ref)]
    print(f"segments {segs}: forward {t_fwd:8.1f} us  image max abs diff {float((img - img1).abs().max()):.2e}  backward {t:8.1f} us   max err / max|ref| (xys conics colors opac) "
          + " ".join(f"{e:.2e}" for e in errs) + "   L2 rel " + " ".join(f"{e:.2e}" for e in l2)
          + f"   image identical {bool((img == img1).all())}")
