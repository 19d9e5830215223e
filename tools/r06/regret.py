#!/usr/bin/env python3
"""The dispatch rule on scene families it was NOT fitted to: for room / floaters / needles
(harness.scene.make_heldout_scene) x {480x270, 960x540, 1080p, 4K}, the raster step (forward + backward through the
public autograd API, bench.py's step) under the DEFAULT tuning table against single-row alternatives
(rasterizer/cuda/_tuning.py overrides) -> a regret table: (default - best) / best per cell.
    python tools/r06/regret.py [out.txt]"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from harness import scene as S  # noqa: E402
from harness.pipeline import CameraTensors, render_view  # noqa: E402
from rasterizer.cuda import _tuning as T  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
CELLS = [(480, 270, 300_000), (960, 540, 500_000), (1920, 1080, 1_000_000), (3840, 2160, 1_000_000)]
if os.environ.get("REGRET_SMALL"):  # the split-all grids only
    CELLS = [(480, 270, 300_000), (400, 300, 150_000)]
if os.environ.get("REGRET_QUICK"):
    CELLS = [(480, 270, 100_000), (1920, 1080, 200_000)]
ALTS = [("default", {}),
        ("deep_factor 0.8", {"deep_factor": 0.8}), ("deep_factor 2.0", {"deep_factor": 2.0}),
        ("deep_factor_bwd 1.0", {"deep_factor_bwd": 1.0}), ("deep_factor_bwd 4.0", {"deep_factor_bwd": 4.0}),
        ("deep_order 0", {"deep_order": 0}), ("deep_tail 0", {"deep_tail": 0}),
        ("depth_segments 1", {"depth_segments": 1}), ("depth_segments 8", {"depth_segments": 8}),
        ("depth_segments_fwd 8", {"depth_segments_fwd": 8}), ("deep_min 1024", {"deep_min": 1024})]
families = sys.argv[2].split(",") if len(sys.argv) > 2 else list(S.HELDOUT_KINDS)
rows = []
for fam in families:
    for (W, H, N) in CELLS:
        cam = S.make_camera(W, H)
        sc = S.make_heldout_scene(fam, N, cam, sh_degree=3)
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        params = {k: t(v).requires_grad_(True) for k, v in sc.items()}
        plist = list(params.values())
        camt = CameraTensors.from_numpy(cam, dev)
        bg = t(np.array(S.BACKGROUND, np.float32))
        v_img_np, v_alpha_np = S.make_cotangents(cam)
        v_img, v_alpha = t(v_img_np), t(v_alpha_np)

        def step():
            for p in plist:
                p.grad = None
            out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"],
                              params["sh_coeffs"], camt, bg, 3, clamp_rgb=False)
            torch.autograd.backward([out["rgb"], out["alpha"]], [v_img, v_alpha[..., None]])
            return out

        cell = {}
        for name, over in ALTS:
            small = ((W + 15) // 16) * ((H + 15) // 16) <= 1100
            if name.startswith("depth_segments") and not small:
                continue  # (rows that only act on split-all grids)
            T.set_overrides(over)
            for _ in range(15):
                out = step()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(30):
                    step()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
            cell[name] = round(best, 4)
        T.set_overrides()
        from rasterizer import rasterize as R
        lens = (R._bin_cache["value"][2][:, 1] - R._bin_cache["value"][2][:, 0]).float()
        best_name = min(cell, key=cell.get)
        regret = cell["default"] / cell[best_name] - 1.0
        rows.append({"family": fam, "res": f"{W}x{H}", "N": N, "mean_list": round(float(lens.mean()), 1),
                     "max_list": int(lens.max()), "ms": cell, "best": best_name, "regret": round(regret, 4)})
        print(json.dumps(rows[-1]), flush=True)
        del params, plist, out
        torch.cuda.empty_cache()
out_path = sys.argv[1] if len(sys.argv) > 1 else None
lines = ["# default tuning table vs single-row alternatives on held-out scene families (tools/r06/regret.py); ms per raster step "
         "(fwd + bwd through the autograd API), best of 3 x 30 steps, one process, 1 x MI355X",
         "family    res        N        mean/max list   default   best (setting)                regret"]
for r in rows:
    lines.append("%-9s %-10s %-8d %6.0f/%-6d   %8.4f  %8.4f (%-22s) %6.1f %%" % (
        r["family"], r["res"], r["N"], r["mean_list"], r["max_list"], r["ms"]["default"], r["ms"][r["best"]], r["best"], 100 * r["regret"]))
lines.append("max regret %.1f %%" % (100 * max(r["regret"] for r in rows)))
lines.append("")
for r in rows:
    lines.append(json.dumps(r))
txt = "\n".join(lines)
print(txt)
if out_path:
    open(out_path, "w").write(txt + "\n")
