"""Forward only (the viewer / evaluation path: `get_outputs` under `torch.no_grad()`):
frames per second through the public `rasterizer` API on the bench scene.

    python tools/render_bench.py [--gaussians N] [--width W] [--height H] [--frames K] [--depth]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch

from harness import scene as S
from harness.pipeline import CameraTensors, render_view


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--scale-lo", type=float, default=0.0025)
    ap.add_argument("--scale-hi", type=float, default=0.025)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--depth", action="store_true", help="also the depth image (one fused compositing pass)")
    ap.add_argument("--fused", action="store_true",
                    help="gs_fused.render_gaussians under no_grad: the whole view as ~10 native calls instead of "
                         "the models' op-by-op sequence (get_outputs_for_camera, vanilla_gs.py:949-962)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cams = [S.make_camera(a.width, a.height, yaw=0.01 * k) for k in range(8)]  # a slowly turning camera
    sc = S.make_scene(a.gaussians, cams[0], sh_degree=a.sh_degree, seed=42, scale_lo=a.scale_lo, scale_hi=a.scale_hi)
    p = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
    camt = [CameraTensors.from_numpy(c, dev) for c in cams]
    bg = torch.tensor(S.BACKGROUND, device=dev)

    if a.fused:
        from gs_fused import ListCapacity, ViewSpec, render_gaussians

        raw = {"means": p["means3d"], "scales": p["scales"].log(), "quats": p["quats"],
               "opacities": torch.logit(p["opacities"]), "features_dc": p["sh_coeffs"][:, 0, :].contiguous(),
               "features_rest": p["sh_coeffs"][:, 1:, :].contiguous()}
        c0 = camt[0]
        spec = ViewSpec(a.height, a.width, c0.fx, c0.fy, c0.cx, c0.cy, a.sh_degree, render_depth=a.depth)
        caps = ListCapacity()
        with torch.no_grad():
            probe = render_gaussians(raw["means"], raw["scales"], raw["quats"], raw["opacities"], raw["features_dc"],
                                     raw["features_rest"], c0.viewmat, c0.projmat, c0.campos, bg, spec, caps.capacity)
            caps.capacity = ((int(1.3 * int(probe["count"].item())) + (1 << 20)) >> 20) << 20

    def frame(k):
        if a.fused:
            c = camt[k % 8]
            with torch.no_grad():
                return render_gaussians(raw["means"], raw["scales"], raw["quats"], raw["opacities"], raw["features_dc"],
                                        raw["features_rest"], c.viewmat, c.projmat, c.campos, bg, spec, caps.capacity)
        with torch.no_grad():
            return render_view(p["means3d"], p["scales"], p["quats"], p["opacities"], p["sh_coeffs"], camt[k % 8], bg,
                               a.sh_degree, render_depth=a.depth, fused_depth=a.depth)

    for k in range(20):
        out = frame(k)
    import gc

    gc.collect()  # (a full collection falling into a 100-ms timed loop costs 0.15 ms per frame: tools/exp/gc_frames.py)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(a.frames):
        out = frame(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.frames
    print(json.dumps({"metric": "forward-only frames/s (no_grad, public API)", "value": round(1.0 / dt, 1),
                      "ms_per_frame": round(dt * 1e3, 4), "mpix_per_s": round(a.width * a.height / dt / 1e6, 1),
                      "gaussians": a.gaussians, "resolution": f"{a.width}x{a.height}", "depth": a.depth, "path": "gs_fused.render_gaussians" if a.fused else "public ops (render_view)",
                      "checksum": float(out["rgb"].double().sum())}))


if __name__ == "__main__":
    main()
