"""Fuzz of the refinement compaction (csrc/refine.hip through gs_fused.refine_gaussians) against the numpy oracle
with RANDOM configurations: thresholds, schedule (warm-up, refine / reset intervals, stop steps), split sample
counts, model sizes, SH degrees, with and without optimizer state and statistics.
python tools/exp/fuzz_refine.py [cases] [seed]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"), os.path.join(ROOT, "tests")]
import numpy as np

spec = importlib.util.spec_from_file_location("tr", os.path.join(ROOT, "tests", "test_refine.py"))
tr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tr)
RO, NAMES = tr.RO, tr.NAMES


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for k in range(cases):
        n = int(rng.choice([1, 2, 255, 256, 257, 5000, 70_001, 200_000]))
        K = int(rng.choice([1, 4, 9, 16]))
        cfg = RO.RefineConfig(
            warmup_length=int(rng.choice([0, 500, 2000])), refine_every=int(rng.choice([50, 100, 300])),
            cull_alpha_thresh=float(rng.choice([0.005, 0.1, 0.5])), cull_scale_thresh=float(rng.choice([0.1, 0.5, 2.0])),
            continue_cull_post_densification=bool(rng.integers(2)), reset_alpha_every=int(rng.choice([3, 30])),
            densify_grad_thresh=float(rng.choice([1e-6, 2e-4, 1e-2])), densify_size_thresh=float(rng.choice([0.003, 0.01, 0.1])),
            n_split_samples=int(rng.choice([1, 2, 3, 4])), cull_screen_size=float(rng.choice([0.05, 0.15])),
            split_screen_size=float(rng.choice([0.01, 0.05])), stop_screen_size_at=int(rng.choice([1000, 4000])),
            stop_split_at=int(rng.choice([3000, 10_000])))
        step = int(rng.choice([1, 400, 700, 1500, 3700, 4500, 9000, 12_000, 16_000]))
        ntd = int(rng.choice([1, 20, 300]))
        max_dim = int(rng.choice([256, 1920, 3840]))
        params = tr._random_model(rng, n, K=max(K, 2)) if K > 1 else tr._random_model(rng, n, K=2)
        if K == 1:
            params["features_rest"] = params["features_rest"][:, :0]
        with_mom, with_stats = bool(rng.integers(4)), bool(rng.integers(5))
        moments = {q: (rng.standard_normal(v.shape).astype(np.float32), rng.uniform(0, 1, v.shape).astype(np.float32))
                   for q, v in params.items()} if with_mom else None
        stats = (np.abs(rng.standard_normal(n) * 10 ** rng.uniform(-8, -3, n)).astype(np.float32),
                 rng.integers(1, 30, n).astype(np.float32), rng.uniform(0, 0.3, n).astype(np.float32)) if with_stats else None
        ok = tr._stable(params, stats, cfg, max_dim)
        params["scales"][~ok] = np.log(0.2 * cfg.cull_scale_thresh + 0.3 * cfg.densify_size_thresh)
        params["opacities"][~ok] = 9.0
        if stats is not None:
            stats[0][~ok] = 0
            stats[2][~ok] = 0
        tag = f"case {k}: n={n} K={K} step={step} S={cfg.n_split_samples} moments={with_mom} stats={with_stats}"
        if not tr._stable(params, stats, cfg, max_dim).all():
            print(tag, "skipped (borderline)")
            continue
        try:
            seed = int(rng.integers(1, 1 << 40))
            try:
                ref_p, ref_m, info = RO.refine(params, moments, stats, cfg, step, ntd, max_dim, samples=None, seed=seed)
            except TypeError:  # densification without statistics: the reference asserts, the product raises
                assert stats is None
                try:
                    tr._gpu_refine(params, moments, stats, cfg, step, ntd, max_dim, samples=None, seed=seed)
                    raise AssertionError("the product accepted a densification step without statistics")
                except ValueError:
                    print(tag, "-> both refuse ok", flush=True)
                    continue
            p, m, ginfo = tr._gpu_refine(params, moments, stats, cfg, step, ntd, max_dim, samples=None, seed=seed)
            for q in NAMES:
                assert p[q].shape == ref_p[q].shape, (q, p[q].shape, ref_p[q].shape)
                if q == "means":
                    np.testing.assert_allclose(p[q], ref_p[q], rtol=1e-5, atol=2e-5)
                elif q == "scales":
                    np.testing.assert_allclose(p[q], ref_p[q], rtol=3e-6, atol=3e-6)
                else:
                    np.testing.assert_array_equal(p[q], ref_p[q])
                if moments is not None:
                    np.testing.assert_array_equal(m[q][0], ref_m[q][0])
                    np.testing.assert_array_equal(m[q][1], ref_m[q][1])
            assert ginfo["n_out"] == ref_p["means"].shape[0]
            print(tag, f"-> {ginfo['n_out']} ok", flush=True)
        except AssertionError as e:
            bad += 1
            print(tag, "MISMATCH", str(e)[:300].replace("\n", " "), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
