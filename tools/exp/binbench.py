#!/usr/bin/env python3
"""Times the binning stages alone on the bench scene (no compositing), so that
experimental / debug variants of the list-building kernels can be timed even when
their output is not valid.  python tools/exp/binbench.py [n] [iters]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import rasterizer.cuda as C  # noqa: E402
from harness import scene as S  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lo, hi = float(os.environ.get("SCALE_LO", 0.0025)), float(os.environ.get("SCALE_HI", 0.025))
    W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if os.environ.get("BLOB"):  # the trainer harness' scene: a ball of Gaussians seen from an orbit camera
        from harness import train as T

        cam = T.orbit_cameras(16, W, H)[3]
        raw = T.blob_scene(n, seed=0, sh_degree=0)
        q = raw["quats"] / np.linalg.norm(raw["quats"], axis=-1, keepdims=True)
        sc = {"means3d": raw["means"], "scales": np.exp(raw["scales"]), "quats": q.astype(np.float32),
              "opacities": (1 / (1 + np.exp(-raw["opacities"]))).astype(np.float32)}
    else:
        cam = S.make_camera(W, H)
        sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=lo, scale_hi=hi)
    cov3d, xys, depths, radii, conics, comp, tiles = C.project_gaussians_forward(
        n, cu(sc["means3d"]), cu(sc["scales"]), 1.0, cu(sc["quats"]), cu(cam.viewmat[:3]), cu(cam.projmat),
        cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16, 0.01)
    opac = cu(sc["opacities"])
    if os.environ.get("SORTED"):  # bound of what records in depth order could buy: order == identity
        perm = torch.argsort(torch.where(radii > 0, depths, torch.full_like(depths, -1.0)), stable=True)
        xys, depths, radii, conics, opac, tiles = (t[perm].contiguous() for t in (xys, depths, radii, conics, opac, tiles))
    print("reference intersections", int(tiles.sum().item()), "visible", int((radii > 0).sum().item()))
    tb = ((W + 15) // 16, (H + 15) // 16, 1)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            out = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3, out

    nb = C.tile_bands(tb) if os.environ.get("BANDED") else 1
    t_cnt, (cnt, recs) = timed(lambda: C.count_reach(xys, radii, conics, opac, tb, bands=nb))
    t_ord, (order, cum) = timed(lambda: C.depth_order(depths, radii, cnt))
    I = int(cum[-1].item())
    t_bin, _ = timed(lambda: C.bin_sorted(n, I, order, cum, xys, radii, tb, 16, recs))
    if os.environ.get("LEAN"):  # lists without counts (two-level partition only)
        cap = int(1.3 * I)
        assert not C.lists_need_counts(n, cap, tb), "these sizes take the single-pass path"
        t_cnt, (_, recs) = timed(lambda: C.count_reach(xys, radii, conics, opac, tb, counts=False))
        t_ord, (order, _) = timed(lambda: C.depth_order(depths, radii, None))
        t_bin, _ = timed(lambda: C.bin_sorted(n, cap, order, None, xys, radii, tb, 16, recs, device_sized=True))
    print(f"n={n} I'={I} count_reach {t_cnt:.1f} us  depth_order {t_ord:.1f} us  bin_sorted {t_bin:.1f} us "
          f"(GSR_TILE_SORT={os.environ.get('GSR_TILE_SORT', '-')}, GSR_DBG={os.environ.get('GSR_DBG', '-')})")


if __name__ == "__main__":
    main()
