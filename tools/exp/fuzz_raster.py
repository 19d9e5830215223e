"""Fuzz of the compositing kernels (16x16 tiles incl. the _ex entry points, and the generic kernels for other
block widths) against the CPU oracle on random image shapes, splat sizes and opacities, with the tests' own
tolerances.  python tools/exp/fuzz_raster.py [cases] [seed]"""
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
O, cu, npy = tk.O, tk.cu, tk.npy
import rasterizer.cuda as C


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for k in range(cases):
        n = int(rng.choice([1, 2, 63, 64, 65, 300, 1500, 6000]))
        W, H = int(rng.integers(1, 330)), int(rng.integers(1, 230))
        bw = int(rng.choice([16, 16, 16, 8, 5, 2]))
        hi = float(rng.choice([0.01, 0.05, 0.2, 1.0]))
        lo = hi * float(rng.choice([0.1, 1.0]))
        ck = dict(yaw=float(rng.uniform(-0.3, 0.3)), pitch=float(rng.uniform(-0.2, 0.2)))
        seed = int(rng.integers(1 << 20))
        tag = f"case {k}: n={n} {W}x{H} bw={bw} scales [{lo:.3f},{hi:.3f}] seed={seed}"
        try:
            d = tk.raster_inputs(n, W, H, bw, ck, seed=seed, scale_lo=lo, scale_hi=hi)
            if d["I"] < 1:
                continue
            opac = (d["opac"] * float(rng.choice([1.0, 1.0, 0.05]))).astype(np.float32)
            if rng.random() < 0.3:
                opac[rng.random(n) < 0.3] = 1.0
            ref = O.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), d["vs"], d["bins"], d["xys"], d["conics"],
                                      d["colors"], opac, d["bg"], ambig_eps=1e-5)
            args = (d["tb"], (bw, bw, 1), (W, H, 1), cu(d["vs"]), cu(d["bins"]), cu(d["xys"]), cu(d["conics"]),
                    cu(d["colors"]), cu(opac), cu(d["bg"]))
            if bw == 16:
                acc = C.backward_accumulators(n, 3, "cuda:0")
                acc.fill_(float("nan"))
                out, Ts, idx, alpha = C.rasterize_forward_ex(*args, want_alpha=True, zero=acc)
                assert torch.equal(alpha, 1 - Ts) and bool((acc == 0).all())
            else:
                acc = None
                out, Ts, idx = C.rasterize_forward(*args)
            ok = ~ref[3]
            if ok.mean() > 0.9:
                np.testing.assert_allclose(npy(out)[ok], ref[0][ok], rtol=0, atol=1e-4)
                np.testing.assert_allclose(npy(Ts)[ok], ref[1][ok], rtol=0, atol=1e-4)
                assert np.array_equal(npy(idx)[ok], ref[2][ok])
            v_img = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
            v_alpha = rng.uniform(-1, 1, (H, W)).astype(np.float32)
            refb = O.rasterize_backward(H, W, bw, d["vs"], d["bins"], d["xys"], d["conics"], d["colors"], opac, d["bg"],
                                        ref[1], ref[2], v_img, v_alpha, with_abs_sums=True, ambig_eps=1e-4)
            sums, amb = refb[4:8], refb[8]  # sum of |per-pixel terms| per component; Gaussians with a borderline pixel
            kw = dict(accumulators=acc) if acc is not None else {}
            got = C.rasterize_backward(H, W, bw, cu(d["vs"]), cu(d["bins"]), cu(d["xys"]), cu(d["conics"]),
                                       cu(d["colors"]), cu(opac), cu(d["bg"]), cu(ref[1]), cu(ref[2]), cu(v_img),
                                       cu(v_alpha), **kw)
            for g, r, a, nm in zip(got, refb[:4], sums, ["v_xy", "v_conic", "v_colors", "v_opacity"]):
                # 1e-3 of the largest gradient + the fp32 error of a sum with cancellation (2e-5 of the terms'
                # magnitudes), on Gaussians none of whose pixels sits on a skip threshold
                g = npy(g).reshape(r.shape)
                err = np.abs(g - r)[~amb]
                tol = (1e-3 * max(1e-6, float(np.abs(r).max())) + 2e-5 * a)[~amb]
                assert (err <= tol).all(), f"{nm}: worst excess {float((err - tol).max()):.3e} (max |grad| {np.abs(r).max():.3e})"
            print(tag, "ok")
        except AssertionError as e:
            bad += 1
            print(tag, "MISMATCH", str(e)[:300].replace("\n", " "))
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
