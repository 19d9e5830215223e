import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from rasterizer import rasterize as R
import rasterizer.cuda as C
from harness import scene as S
from harness.pipeline import CameraTensors, render_view
from harness.train import GaussianParams, blob_scene, orbit_cameras, seed_model

dev = torch.device("cuda", 0)
# 1) a 1 M scene at 1080p leaves its count hint (what tests/test_gpu_fullsize.py does)
cam = S.make_camera(1920, 1080)
sc = S.make_scene(1_000_000, cam, sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
p = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
with torch.no_grad():
    for _ in range(2):
        render_view(p["means3d"], p["scales"], p["quats"], p["opacities"], p["sh_coeffs"], CameraTensors.from_numpy(cam, dev),
                    torch.tensor(S.BACKGROUND, device=dev), 3)
torch.cuda.synchronize(); print("hint", R._count_hint, flush=True)
orig = R._build_two_round
def spy(xys, depths, radii, conics, tiles, opacity, tb, bw, plan, remember, round1):
    print("two-round plan", {k: plan[k] for k in ("n1", "cap1", "cap2", "f")}, "n", xys.shape[0], flush=True)
    out = orig(xys, depths, radii, conics, tiles, opacity, tb, bw, plan, remember, round1)
    torch.cuda.synchronize()
    ids, bins1 = out[1], out[2]
    aux = R.last_list_aux()
    print("  built: bins1 max", int(bins1.max()), "bins2 max", int(aux[1].max()), "base", aux[2], "ids numel", ids.numel(), flush=True)
    return out
R._build_two_round = spy
cfg = bench.config3(10)
cams = [CameraTensors.from_numpy(c, dev) for c in orbit_cameras(cfg.num_views, 1920, 1080, radius=cfg.cam_radius)]
bg = torch.tensor(S.BACKGROUND, device=dev)
raw_truth = blob_scene(cfg.num_gaussians, seed=cfg.seed, sh_degree=3, kind=cfg.scene, scale_lo=cfg.scene_scale[0],
                       scale_hi=cfg.scene_scale[1], tex_cell=cfg.tex_cell, objects=cfg.scene_objects, extent=cfg.scene_extent)
truth = GaussianParams(raw_truth, dev)
with torch.no_grad():
    for i in range(6):
        truth.render(cams[i], bg, 3)
        torch.cuda.synchronize(); print("truth view", i, "ok", R._two_hint, flush=True)
if os.environ.get("DBG_MODEL_SINGLE"):
    os.environ["GSR_TWO_ROUND"] = "0"
import faulthandler; faulthandler.enable()
origf = R._build_fresh
def spyf(*a, **k):
    print("build_fresh n", a[0].shape[0], "speculate", k.get("speculate", True), "round1", k.get("round1") is not None, "hint", R._count_hint, flush=True)
    return origf(*a, **k)
R._build_fresh = spyf
for nm in ("count_reach", "depth_order", "bin_sorted", "tile_lists_subrange", "saturation_filter", "rasterize_forward_round", "rasterize_forward_ex"):
    def mk(nm, fn):
        def w(*a, **k):
            print("  call", nm, flush=True)
            r = fn(*a, **k)
            torch.cuda.synchronize()
            return r
        return w
    setattr(C, nm, mk(nm, getattr(C, nm)))
model = GaussianParams(seed_model(raw_truth, 200_000, "sfm", 1, 3), dev)
with torch.no_grad():
    for i in range(4):
        model.render(cams[i * 10], bg, 3)
        torch.cuda.synchronize(); print("model view", i, "ok", flush=True)
