"""Refinement (densify / cull / split / duplicate + Adam-state surgery), SURVEY 8f row f1.

CPU: the numpy oracle (`oracle/refine.py`) against the goldens produced by the
reference's own methods (`tests/golden/make_golden_refine.py`).
GPU: `gs_fused.refine_gaussians` (csrc/refine.hip through the C ABI) against the
goldens and against the oracle on larger random models.
"""
import os

import numpy as np
import pytest

from oracle import refine as RO

NAMES = RO.PARAM_NAMES


def load_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "refine.npz"))
    out = {}
    for name in z["cases"]:
        pre = f"{name}/"
        out[str(name)] = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    return out


def case_config(c) -> RO.RefineConfig:
    kw = {}
    for f in RO.RefineConfig.__dataclass_fields__:
        v = c["cfg_" + f]
        kw[f] = type(getattr(RO.RefineConfig(), f))(v)
    return RO.RefineConfig(**kw)


def case_stats(c):
    """Densification statistics of the case, rebuilt by the oracle's `update_stats`
    from the per-view inputs (and checked against what the reference's after_train held)."""
    stats = None
    size = tuple(int(x) for x in c["size"])
    for i in range(int(c["n_views"])):
        stats = RO.update_stats(stats, c[f"view{i}_vxy"], c[f"view{i}_radii"], max(size))
    return stats


CASES = ["warmup", "densify_screen", "densify_bigcull", "densify_late", "densify_3samples", "no_densify_window",
         "opacity_reset", "cull_only", "cull_off", "densify_lowthresh"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_refinement(golden_dir, name):
    c = load_cases(golden_dir)[name]
    cfg = case_config(c)
    stats = case_stats(c)
    if stats is not None:
        # after_train: the statistics the reference accumulated (torch's norm rounds differently
        # from sqrt(x^2 + y^2) by an ulp); the refinement below starts from the reference's own
        np.testing.assert_allclose(stats[0], c["stat_gn"], rtol=5e-7, atol=0)
        np.testing.assert_array_equal(stats[1], c["stat_vc"])
        np.testing.assert_array_equal(stats[2], c["stat_m2"])
        stats = (c["stat_gn"], c["stat_vc"], c["stat_m2"])
    params = {k: c["in_" + k] for k in NAMES}
    moments = {k: (c["in_m_" + k], c["in_v_" + k]) for k in NAMES}
    p, mom, info = RO.refine(params, moments, stats, cfg, int(c["step"]), int(c["num_train_data"]),
                             int(max(c["size"])), samples=c["samples"] if len(c["samples"]) else None)
    for k in NAMES:
        assert p[k].shape == c["out_" + k].shape, (k, p[k].shape, c["out_" + k].shape)
        if k in ("means", "scales"):
            # computed values (exp / log / rotation): a few ulp between numpy and torch
            np.testing.assert_allclose(p[k], c["out_" + k], rtol=2e-6, atol=2e-6)
        else:
            np.testing.assert_array_equal(p[k], c["out_" + k])  # moved values: exact
        np.testing.assert_array_equal(mom[k][0], c["out_m_" + k])
        np.testing.assert_array_equal(mom[k][1], c["out_v_" + k])


def test_counter_based_normals_are_standard_normal():
    i = np.arange(200_000)
    z = np.concatenate([RO.split_normals(7, i, j) for j in range(2)], 0)
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    # independent of how many other Gaussians are asked for, different per seed / sample
    np.testing.assert_array_equal(RO.split_normals(7, np.array([5, 17]), 1), RO.split_normals(7, i, 1)[[5, 17]])
    assert not np.array_equal(RO.split_normals(8, i[:100], 0), RO.split_normals(7, i[:100], 0))
    c = np.corrcoef(z[:, 0], z[:, 1])[0, 1]
    assert abs(c) < 1e-2


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10 (kat_vectors: zero and all-ones inputs)
    z = RO.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros((1, 2), np.uint32))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    o = RO.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), np.full((1, 2), 0xFFFFFFFF, np.uint32))[0]
    assert [hex(int(v)) for v in o] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


# ---- GPU: csrc/refine.hip through the C ABI ------------------------------------------
def _gpu_refine(c_params, c_moments, stats, cfg, step, ntd, max_dim, samples=None, seed=0):
    import torch

    from gs_fused import RefineConfig, refine_gaussians

    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    params = {k: t(v) for k, v in c_params.items()}
    moments = None if c_moments is None else {k: (t(a), t(b)) for k, (a, b) in c_moments.items()}
    st = None
    if stats is not None:
        st = (t(stats[0]), t(stats[1].astype(np.int32)), t(stats[2]))
    gcfg = RefineConfig(**{f: getattr(cfg, f) for f in RefineConfig.__dataclass_fields__})
    p, m, info = refine_gaussians(params, moments, st, gcfg, step, ntd, max_dim,
                                  samples=None if samples is None else t(samples), seed=seed)
    torch.cuda.synchronize()
    p = {k: v.cpu().numpy() for k, v in p.items()}
    m = None if m is None else {k: (a.cpu().numpy(), b.cpu().numpy()) for k, (a, b) in m.items()}
    return p, m, info


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_refinement_matches_reference_goldens(golden_dir, name):
    """The HIP compaction reproduces what the reference's own refinement_after did:
    same rows in the same order (bit-exact for everything that is moved, Adam moments
    included), computed values (split means, shrunk scales) to fp32 rounding."""
    c = load_cases(golden_dir)[name]
    cfg = case_config(c)
    stats = (c["stat_gn"], c["stat_vc"], c["stat_m2"]) if "stat_gn" in c else None
    params = {k: c["in_" + k] for k in NAMES}
    moments = {k: (c["in_m_" + k], c["in_v_" + k]) for k in NAMES}
    p, mom, info = _gpu_refine(params, moments, stats, cfg, int(c["step"]), int(c["num_train_data"]),
                               int(max(c["size"])), samples=c["samples"] if len(c["samples"]) else None)
    for k in NAMES:
        assert p[k].shape == c["out_" + k].shape, (k, p[k].shape, c["out_" + k].shape)
        if k in ("means", "scales"):
            np.testing.assert_allclose(p[k], c["out_" + k], rtol=3e-6, atol=3e-6)
        else:
            np.testing.assert_array_equal(p[k], c["out_" + k])
        np.testing.assert_array_equal(mom[k][0], c["out_m_" + k])
        np.testing.assert_array_equal(mom[k][1], c["out_v_" + k])


def _random_model(rng, n, K=16):
    f = np.float32
    return {
        "means": rng.standard_normal((n, 3)).astype(f),
        "scales": rng.uniform(np.log(0.002), np.log(0.9), (n, 3)).astype(f),
        "quats": rng.standard_normal((n, 4)).astype(f),
        "features_dc": rng.standard_normal((n, 3)).astype(f),
        "features_rest": rng.standard_normal((n, K - 1, 3)).astype(f),
        "opacities": rng.uniform(-4.0, 4.0, (n, 1)).astype(f),
    }


def _stable(params, stats, cfg, max_dim, eps=1e-5):
    """Gaussians none of whose threshold comparisons is within `eps` (relative) of
    flipping: expf / logf differ by an ulp between the device and numpy."""
    e = np.exp(params["scales"].astype(np.float64))
    emax, eshr = e.max(-1), (e / 1.6).max(-1)
    near = lambda v, t: np.abs(v - t) <= eps * abs(t)
    bad = near(emax, cfg.densify_size_thresh) | near(eshr, cfg.densify_size_thresh)
    bad |= near(emax, cfg.cull_scale_thresh) | near(eshr, cfg.cull_scale_thresh)
    bad |= near(1 / (1 + np.exp(-params["opacities"].reshape(-1).astype(np.float64))), cfg.cull_alpha_thresh)
    if stats is not None:
        gn, vc, m2 = stats
        bad |= near(gn.astype(np.float64) / vc * 0.5 * max_dim, cfg.densify_grad_thresh)
        bad |= near(m2, cfg.split_screen_size) | near(m2, cfg.cull_screen_size)
    return ~bad


@pytest.mark.gpu
@pytest.mark.parametrize("n,step,S", [(50_000, 700, 2), (200_000, 3700, 2), (70_001, 4500, 3), (1, 700, 2),
                                      (257, 12_000, 2), (300_000, 12_000, 2)])
def test_hip_refinement_matches_oracle(n, step, S):
    """Larger random models against the numpy oracle, with samples handed in and with
    the in-kernel counter-based generator (same definition in oracle/refine.py)."""
    rng = np.random.default_rng(n + step)
    cfg = RO.RefineConfig(n_split_samples=S)
    params = _random_model(rng, n)
    moments = {k: (rng.standard_normal(v.shape).astype(np.float32),
                   rng.uniform(0, 1, v.shape).astype(np.float32)) for k, v in params.items()}
    max_dim = 1920
    stats = (np.abs(rng.standard_normal(n) * 10 ** rng.uniform(-8, -5, n)).astype(np.float32),
             rng.integers(1, 30, n).astype(np.float32), rng.uniform(0, 0.2, n).astype(np.float32))
    # make every decision numerically stable (nudge the few borderline Gaussians away)
    ok = _stable(params, stats, cfg, max_dim)
    params["scales"][~ok] = np.log(0.2)
    params["opacities"][~ok] = 1.0
    stats[0][~ok] = 0
    stats[2][~ok] = 0
    assert _stable(params, stats, cfg, max_dim).all()
    for seed in (None, 1234567890123):
        ref_p, ref_m, info = RO.refine(params, moments, stats, cfg, step, 20, max_dim, samples=None,
                                       seed=seed or 0)
        samples = None
        if seed is None and info["samples"] is not None:
            # hand the oracle's draws in, in the reference's layout
            samples = info["samples"]
        p, m, ginfo = _gpu_refine(params, moments, stats, cfg, step, 20, max_dim, samples=samples, seed=seed or 0)
        for k in NAMES:
            assert p[k].shape == ref_p[k].shape, (k, p[k].shape, ref_p[k].shape)
            if k == "means":
                # the in-kernel normals go through logf / sqrtf / cosf / sinf: a few ulp of a value ~ 3 sigma * scale
                np.testing.assert_allclose(p[k], ref_p[k], rtol=1e-5, atol=2e-5)
            elif k == "scales":
                np.testing.assert_allclose(p[k], ref_p[k], rtol=3e-6, atol=3e-6)
            else:
                np.testing.assert_array_equal(p[k], ref_p[k])
            np.testing.assert_array_equal(m[k][0], ref_m[k][0])
            np.testing.assert_array_equal(m[k][1], ref_m[k][1])
        assert ginfo["n_out"] == ref_p["means"].shape[0]


@pytest.mark.gpu
def test_hip_refinement_without_optimizer_state_and_everything_culled():
    rng = np.random.default_rng(5)
    cfg = RO.RefineConfig()
    params = _random_model(rng, 1000)
    params["opacities"][:] = -6.0  # sigmoid = 0.0025 < 0.1: every Gaussian is culled
    p, m, info = _gpu_refine(params, None, None, cfg, 12_000, 20, 100)
    assert m is None and info["n_out"] == 0
    for k in NAMES:
        assert p[k].shape == (0,) + params[k].shape[1:]
    # nothing to do: the very same tensors come back
    import torch

    from gs_fused import RefineConfig, refine_gaussians

    params = _random_model(rng, 1000)
    params["opacities"][:] = 3.0
    params["scales"][:] = np.log(0.05)
    tp = {k: torch.from_numpy(v).cuda() for k, v in params.items()}
    out, _, info = refine_gaussians(tp, None, None, RefineConfig(), 12_000, 20, 100)
    assert all(out[k] is tp[k] for k in NAMES) and info["n_out"] == 1000


@pytest.mark.gpu
def test_densify_stats_first_call_semantics():
    """`first=True` = the reference's first after_train after a refinement
    (vanilla_gs.py:354-356): everything starts with count 1 and its own gradient."""
    import torch

    from gs_fused import densify_stats_

    rng = np.random.default_rng(3)
    n = 10_000
    stats = None
    dev = "cuda:0"
    gn = torch.full((n,), float("nan"), device=dev)
    vc = torch.full((n,), -7, dtype=torch.int32, device=dev)
    m2 = torch.full((n,), float("nan"), device=dev)
    for it in range(3):
        radii = ((rng.uniform(0, 1, n) < 0.6) * rng.integers(1, 300, n)).astype(np.int32)
        vxy = (rng.standard_normal((n, 2)) * 1e-4).astype(np.float32)
        vxy[radii == 0] = 0
        stats = RO.update_stats(stats, vxy, radii, 1920)
        densify_stats_(torch.from_numpy(vxy).to(dev), torch.from_numpy(radii).to(dev), 1920, gn, vc, m2,
                       first=(it == 0))
    np.testing.assert_allclose(gn.cpu().numpy(), stats[0], rtol=1e-6)
    np.testing.assert_array_equal(vc.cpu().numpy(), stats[1].astype(np.int32))
    np.testing.assert_allclose(m2.cpu().numpy(), stats[2], rtol=1e-6)


@pytest.mark.gpu
def test_hip_refinement_of_an_sh_degree_0_model():
    """`features_rest` of an SH-degree-0 model is [N, 0, 3]: nothing to move, but the row count must follow
    (found by tools/exp/fuzz_refine.py: the compaction used to refuse the zero-width tensor)."""
    rng = np.random.default_rng(11)
    n = 5000
    params = _random_model(rng, n, K=2)
    params["features_rest"] = params["features_rest"][:, :0]
    moments = {k: (rng.standard_normal(v.shape).astype(np.float32), rng.uniform(0, 1, v.shape).astype(np.float32))
               for k, v in params.items()}
    cfg = RO.RefineConfig()
    stats = (np.abs(rng.standard_normal(n) * 1e-6).astype(np.float32), rng.integers(1, 30, n).astype(np.float32),
             rng.uniform(0, 0.2, n).astype(np.float32))
    ok = _stable(params, stats, cfg, 1920)
    params["scales"][~ok] = np.log(0.2)
    params["opacities"][~ok] = 1.0
    stats[0][~ok] = 0
    stats[2][~ok] = 0
    ref_p, ref_m, _ = RO.refine(params, moments, stats, cfg, 3700, 20, 1920, samples=None, seed=5)
    p, m, info = _gpu_refine(params, moments, stats, cfg, 3700, 20, 1920, samples=None, seed=5)
    assert info["n_out"] == ref_p["means"].shape[0] != n
    assert p["features_rest"].shape == (info["n_out"], 0, 3) and m["features_rest"][0].shape == (info["n_out"], 0, 3)
    for k in ("quats", "features_dc", "opacities"):
        np.testing.assert_array_equal(p[k], ref_p[k])
        np.testing.assert_array_equal(m[k][0], ref_m[k][0])


@pytest.mark.gpu
def test_hip_refinement_with_random_configurations():
    """tools/exp/fuzz_refine.py: 60 random configurations (thresholds, schedule, 1-4 split samples, SH degree
    0-3, 1 to 200 k Gaussians, with and without optimizer state / statistics) against the numpy oracle."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_refine.py"), "60", "51"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") >= 50
