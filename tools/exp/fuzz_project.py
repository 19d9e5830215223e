"""Fuzz of the projection (forward: bit-identical to the oracle; backward: 1e-3 per row) and SH kernels on random
cameras, image shapes, scales from 1e-4 to 30 scene units, points behind / at the near plane, global scale and
clip threshold.  python tools/exp/fuzz_project.py [cases] [seed]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

spec = importlib.util.spec_from_file_location("tk", os.path.join(ROOT, "tests", "test_gpu_kernels.py"))
tk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tk)
O, cu, npy, S = tk.O, tk.cu, tk.npy, tk.S
import rasterizer.cuda as C

NAMES = ["cov3d", "xys", "depths", "radii", "conics", "compensation", "num_tiles_hit"]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for k in range(cases):
        n = int(rng.choice([1, 63, 64, 65, 1000, 20000]))
        W, H = int(rng.integers(1, 2000)), int(rng.integers(1, 1200))
        bw = int(rng.choice([16, 16, 8, 3, 2]))
        hi = float(rng.choice([1e-3, 0.05, 1.0, 30.0]))
        lo = hi * float(rng.choice([0.1, 1.0]))
        ck = dict(yaw=float(rng.uniform(-0.5, 0.5)), pitch=float(rng.uniform(-0.4, 0.4)), roll=float(rng.uniform(-1, 1)),
                  trans=tuple(float(x) for x in rng.uniform(-1, 1, 3)))
        glob, clip = float(rng.choice([1.0, 0.3, 2.5])), float(rng.choice([0.01, 0.5, 3.0]))
        seed = int(rng.integers(1 << 20))
        tag = f"case {k}: n={n} {W}x{H} bw={bw} scales [{lo:.4f},{hi:.4f}] glob={glob} clip={clip} seed={seed}"
        try:
            cam = S.make_camera(W, H, **ck)
            sc = S.make_scene(n, cam, sh_degree=int(rng.integers(0, 4)), seed=seed, scale_lo=lo, scale_hi=hi,
                              z_lo=float(rng.choice([-1.0, 0.005, 2.0])), z_hi=10.0)
            ref = tk.project_cpu(cam, sc, bw, clip=clip, glob=glob)
            out = [npy(t) for t in tk.project_gpu(cam, sc, bw, clip=clip, glob=glob)]
            for nm, o, r in zip(NAMES, out, ref):
                same = np.array_equal(o, r) or np.array_equal(np.nan_to_num(o, nan=-7.0), np.nan_to_num(r, nan=-7.0))
                assert same, f"{nm}: {(o != r).sum()} elements differ"
            cov3d, xys, depths, radii, conics, comp, tiles = ref
            v_xy = rng.standard_normal((n, 2)).astype(np.float32)
            v_depth = rng.standard_normal(n).astype(np.float32)
            v_conic = rng.standard_normal((n, 3)).astype(np.float32)
            v_comp = rng.standard_normal(n).astype(np.float32)
            rb = O.project_gaussians_backward(n, sc["means3d"], sc["scales"], glob, sc["quats"], cam.viewmat[:3],
                                              cam.projmat, cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width,
                                              cov3d, radii, conics, comp, v_xy, v_depth, v_conic, v_comp)
            ob = C.project_gaussians_backward(n, cu(sc["means3d"]), cu(sc["scales"]), glob, cu(sc["quats"]),
                                              cu(cam.viewmat[:3]), cu(cam.projmat), cam.fx, cam.fy, cam.cx, cam.cy,
                                              cam.height, cam.width, cu(cov3d), cu(radii), cu(conics), cu(comp),
                                              cu(v_xy), cu(v_depth), cu(v_conic), cu(v_comp))
            for o, r, nm in zip(ob, rb, ["v_cov2d", "v_cov3d", "v_mean3d", "v_scale", "v_quat"]):
                o = npy(o)
                fin = np.isfinite(r).all(axis=-1) if r.ndim > 1 else np.isfinite(r)
                assert np.all(o[radii <= 0] == 0), nm + " (culled rows)"
                if not fin.any():
                    continue
                r2, o2 = r[fin], o[fin]
                rowmax = np.abs(r2).max(axis=-1, keepdims=True) if r2.ndim > 1 else np.abs(r2)
                e = np.abs(o2 - r2) / np.maximum(rowmax, 1e-6 * max(1e-30, float(np.abs(r2).max())))
                assert e.max() < 1e-3, f"{nm}: {e.max():.3e}"
            # SH forward / backward on the same Gaussians
            deg = {1: 0, 4: 1, 9: 2, 16: 3}[sc["sh_coeffs"].shape[1]]
            use = int(rng.integers(0, deg + 1))
            dirs = S.viewdirs_for(sc, cam)
            rs = O.compute_sh_forward(n, deg, use, dirs, sc["sh_coeffs"])
            os_ = npy(C.compute_sh_forward(n, deg, use, cu(dirs), cu(sc["sh_coeffs"])))
            np.testing.assert_allclose(os_, rs, rtol=1e-5, atol=1e-6)
            vcol = rng.standard_normal((n, 3)).astype(np.float32)
            rsb = O.compute_sh_backward(n, deg, use, dirs, vcol)
            osb = npy(C.compute_sh_backward(n, deg, use, cu(dirs), cu(vcol)))
            np.testing.assert_allclose(osb, rsb, rtol=1e-5, atol=1e-6)
            print(tag, "ok", flush=True)
        except AssertionError as e:
            bad += 1
            print(tag, "MISMATCH", str(e)[:300].replace("\n", " "), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
