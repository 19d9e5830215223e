"""Fuzz of the list builders: the two-level partition (forced: GSR_TILE_SORT=t) against the single-pass /
banded scatter (selected by want_slots) on random sizes, grids and splat sizes; also the count-free
flow.  GSR_TILE_SORT=t python tools/exp/fuzz_lists.py [cases] [seed]"""
import os
import sys

os.environ.setdefault("GSR_TILE_SORT", "t")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch
import rasterizer.cuda as C
from harness import scene as S


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bad = 0
    for k in range(cases):
        n = int(rng.choice([1, 7, 63, 64, 65, 255, 256, 257, 1000, 5000, 20000, 66000, 150000, 400000]))
        W = int(rng.choice([16, 33, 160, 640, 1920, 2560, 3840, 272, 16384]))
        H = int(rng.choice([16, 47, 96, 360, 1080, 1440, 2160, 16368, 272]))
        if W * H > 3840 * 2160 * 2:
            H = 272
        hi = float(rng.choice([0.005, 0.02, 0.08, 0.5, 2.0]))
        lo = hi * float(rng.choice([0.05, 0.3, 1.0]))
        yaw = float(rng.uniform(-0.3, 0.3))
        cam = S.make_camera(W, H, yaw=yaw)
        sseed = int(rng.integers(1 << 30))
        sc = S.make_scene(n, cam, sh_degree=0, seed=sseed, scale_lo=lo, scale_hi=hi)
        ofac = float(rng.choice([1.0, 0.3, 0.01]))
        opac = (sc["opacities"] * ofac).astype(np.float32)
        ostep = int(rng.integers(2, 9)) if rng.random() < 0.3 else 0
        if ostep:
            opac[::ostep] = 0.0
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"  begin case {k}: n={n} {W}x{H} lo={lo} hi={hi} yaw={yaw} seed={sseed} ofac={ofac} ostep={ostep}", flush=True)
        out = C.project_gaussians_forward(n, cu(sc["means3d"]), cu(sc["scales"]), 1.0, cu(sc["quats"]),
                                          cu(cam.viewmat[:3]), cu(cam.projmat), cam.fx, cam.fy, cam.cx, cam.cy,
                                          cam.height, cam.width, 16, 0.01)
        cov3d, xys, depths, radii, conics, comp, tiles = out
        tb = ((W + 15) // 16, (H + 15) // 16, 1)
        if tb[0] > 1024 or tb[1] >= 1024:
            continue
        op = cu(opac)
        nb = C.tile_bands(tb)
        if int(tiles.sum(dtype=torch.int64).item()) >= 2 ** 31 - 2 ** 24:
            continue  # more box intersections than int32 lists hold (the reference's limit as well)
        cnt1, recs = C.count_reach(xys, radii, conics, op, tb)
        order, cum = C.depth_order(depths, radii, cnt1)
        I = int(cum[-1].item()) if n else 0
        if I < 1:
            continue
        ids_t, bins_t = C.bin_sorted(n, I, order, cum, xys, radii, tb, 16, recs)
        cntb, recsb = C.count_reach(xys, radii, conics, op, tb, bands=nb)
        orderb, cumb = C.depth_order(depths, radii, cntb)
        ids_s, bins_s, _ = C.bin_sorted(n, I, orderb, cumb, xys, radii, tb, 16, recsb, want_slots=True)
        ok = torch.equal(ids_t, ids_s) and torch.equal(bins_t, bins_s)
        # count-free flow, capacity above and below the need
        _, recs2 = C.count_reach(xys, radii, conics, op, tb, counts=False)
        order2, _ = C.depth_order(depths, radii, None)
        count = torch.zeros(1, dtype=torch.int32).pin_memory()
        cap = I + int(rng.integers(0, 5000))
        ids_l, bins_l = C.bin_sorted(n, cap, order2, None, xys, radii, tb, 16, recs2, device_sized=True, count_out=count)
        torch.cuda.synchronize()
        ok = ok and int(count[0]) == I and torch.equal(ids_l[:I], ids_s) and torch.equal(bins_l, bins_s)
        small = max(1, I // int(rng.integers(2, 6)))
        ids_c, bins_c = C.bin_sorted(n, small, order2, None, xys, radii, tb, 16, recs2, device_sized=True,
                                     count_out=count)
        torch.cuda.synchronize()
        ok = ok and int(count[0]) == I and int(bins_c.max()) <= small
        print(f"case {k}: n={n} {W}x{H} scales [{lo:.4f},{hi:.4f}] I'={I} bands={nb} {'ok' if ok else 'MISMATCH'}")
        bad += 0 if ok else 1
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
