"""CPU: the optional loss terms of the reference's co-gs model (harness/cogs_losses.py; all off in the reference's default
config, depth_gs.py:93-139).  PINNED (round 6) on tests/golden/cogs_losses.npz: values the reference's OWN code produced
-- `pearson_depth_loss` / `local_pearson_loss` / `tv_Loss` of gs_toolkit/utils/losses.py and the scale-regularisation,
sparse and scaled log-depth blocks of `DepthGSModel.get_loss_dict`, lifted out with `ast` and executed by
tests/golden/make_golden_cogs.py -- and, as before, checked against independent float64 numpy restatements of the
formulas; then the trainer's co-gs loop with the switches on, on the oracle-backed stand-ins of the native ops.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from harness import cogs_losses as CL  # noqa: E402


GOLDEN = os.path.join(ROOT, "tests", "golden", "cogs_losses.npz")


def test_every_optional_term_equals_what_the_references_own_code_produced():
    g = np.load(GOLDEN)
    for k in ("c0_", "c1_", "c2_"):
        pred, gt, img = (torch.from_numpy(g[k + n]) for n in ("pred", "gt", "img"))
        box, p_corr = int(g[k + "box_pcorr"][0]), float(g[k + "box_pcorr"][1])
        assert float(CL.pearson_depth_loss(pred.reshape(-1), gt.reshape(-1))) == pytest.approx(float(g[k + "pearson"]), abs=2e-6)
        corners = (torch.from_numpy(g[k + "patch_rows"]), torch.from_numpy(g[k + "patch_cols"]))
        # the source draws int(p_corr * floor(H / box) * floor(W / box)) corners: so does local_pearson_patches
        assert corners[0].numel() == CL.local_pearson_patches(pred.shape[0], pred.shape[1], box, p_corr)[0].numel()
        got = CL.local_pearson_loss(pred, gt, box, p_corr, corners=corners)
        assert float(got) == pytest.approx(float(g[k + "local_pearson"]), abs=5e-6)
        assert float(CL.tv_loss(pred)) == pytest.approx(float(g[k + "tv"]), rel=2e-6)
        scale, shift = (float(v) for v in g[k + "scale_shift"])
        assert float(CL.scaled_log_depth_loss(pred, gt, img, scale, shift)) == pytest.approx(float(g[k + "log_depth"]), rel=2e-6)
    for k in ("s0_", "s1_"):
        ratio, lam = (float(v) for v in g[k + "ratio_lambda"])
        assert float(CL.scale_regularisation(torch.from_numpy(g[k + "log_scales"]), ratio)) == pytest.approx(float(g[k + "scale_reg"]), rel=2e-6)
        assert float(CL.sparse_loss(torch.from_numpy(g[k + "opacities"]), lam)) == pytest.approx(float(g[k + "sparse_loss"]), rel=2e-6)


def _pearson_np(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    cov = np.mean((a - a.mean()) * (b - b.mean()))          # biased, as the source's torch.mean
    return 1.0 - cov / (a.std(ddof=1) * b.std(ddof=1))      # torch.std is unbiased


def _images(h=40, w=56, seed=0):
    rng = np.random.default_rng(seed)
    gt = rng.uniform(0.5, 4.0, (h, w)).astype(np.float32)
    pred = (0.7 * gt + 0.3 * rng.uniform(0.5, 4.0, (h, w))).astype(np.float32)
    img = rng.uniform(0.0, 1.0, (h, w, 3)).astype(np.float32)
    return pred, gt, img


def test_pearson_follows_the_source_biased_covariance_over_unbiased_deviations():
    pred, gt, _ = _images()
    got = float(CL.pearson_depth_loss(torch.from_numpy(pred).reshape(-1), torch.from_numpy(gt).reshape(-1)))
    assert got == pytest.approx(_pearson_np(pred, gt), abs=2e-6)
    n = pred.size
    r = np.corrcoef(pred.ravel().astype(np.float64), gt.ravel().astype(np.float64))[0, 1]
    assert got == pytest.approx(1.0 - (n - 1) / n * r, abs=2e-6)  # what the mixed normalisation amounts to
    # perfectly correlated inputs do not reach 0 with n samples: 1 - (n - 1) / n
    x = torch.linspace(0.0, 1.0, 10)
    assert float(CL.pearson_depth_loss(x, 3.0 * x + 2.0)) == pytest.approx(0.1, abs=1e-6)


def test_local_pearson_equals_the_sources_loop_over_the_same_patches():
    pred, gt, _ = _images(48, 64, seed=1)
    box, p_corr = 16, 0.5
    g = torch.Generator().manual_seed(7)
    x0, y0 = CL.local_pearson_patches(48, 64, box, p_corr, g)
    assert x0.numel() == int(p_corr * math.floor(48 / box) * math.floor(64 / box)) == 6
    assert int(x0.max()) < 48 - box and int(y0.max()) < 64 - box  # randint's upper bound is exclusive, as in the source
    want = np.mean([_pearson_np(pred[a:a + box, b:b + box], gt[a:a + box, b:b + box])
                    for a, b in zip(x0.tolist(), y0.tolist())])
    got = CL.local_pearson_loss(torch.from_numpy(pred)[..., None], torch.from_numpy(gt), box, p_corr, corners=(x0, y0))
    assert float(got) == pytest.approx(want, abs=5e-6)
    # drawn inside: the same generator state gives the same patches and the same value
    again = CL.local_pearson_loss(torch.from_numpy(pred), torch.from_numpy(gt), box, p_corr,
                                  generator=torch.Generator().manual_seed(7))
    assert float(again) == pytest.approx(float(got), abs=1e-7)
    # a patch of constant depth has no deviation: 0 / 0, as in the source (no epsilon there, none here)
    flat = torch.ones(48, 64)
    assert math.isnan(float(CL.local_pearson_loss(flat, torch.from_numpy(gt), box, p_corr, corners=(x0, y0))))
    # differentiable with respect to the rendered depth
    p = torch.from_numpy(pred).clone().requires_grad_(True)
    CL.local_pearson_loss(p, torch.from_numpy(gt), box, p_corr, corners=(x0, y0)).backward()
    assert torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0


def test_tv_and_scaled_log_depth_terms():
    pred, gt, img = _images(seed=2)
    p64, g64, i64 = pred.astype(np.float64), gt.astype(np.float64), img.astype(np.float64)
    tv = np.abs(p64[:, :-1] - p64[:, 1:]).mean() + np.abs(p64[:-1] - p64[1:]).mean()
    assert float(CL.tv_loss(torch.from_numpy(pred))) == pytest.approx(tv, rel=2e-6)
    scale, shift = 1.3, -0.2
    logl1 = np.log(1 + np.abs(g64 - (scale * p64 + shift)))
    lx = np.exp(-np.abs(i64[:, :-1] - i64[:, 1:]).mean(-1)) * logl1[:, :-1]
    ly = np.exp(-np.abs(i64[:-1] - i64[1:]).mean(-1)) * logl1[:-1]
    got = CL.scaled_log_depth_loss(torch.from_numpy(pred)[..., None], torch.from_numpy(gt), torch.from_numpy(img), scale, shift)
    assert float(got) == pytest.approx(lx.mean() + ly.mean(), rel=2e-6)


def test_scale_regularisation_and_sparse_loss():
    rng = np.random.default_rng(3)
    log_s = rng.normal(-3.0, 1.2, (500, 3)).astype(np.float32)
    s = np.exp(log_s.astype(np.float64))
    ratio = s.max(-1) / s.min(-1)
    want = 0.1 * (np.maximum(ratio, 10.0) - 10.0).mean()
    assert (ratio > 10.0).any() and (ratio < 10.0).any()
    assert float(CL.scale_regularisation(torch.from_numpy(log_s), 10.0)) == pytest.approx(want, rel=1e-5)
    assert float(CL.scale_regularisation(torch.zeros(4, 3), 10.0)) == 0.0  # round Gaussians pay nothing
    o = rng.uniform(0.05, 0.95, 300).astype(np.float32)
    want = 0.1 * (np.log(o.astype(np.float64) + 1e-6) + np.log(1 - o.astype(np.float64) + 1e-6)).mean()
    assert float(CL.sparse_loss(torch.from_numpy(o), 0.1)) == pytest.approx(want, rel=1e-5)
    # the source hands the raw parameter (a logit) to the logs: outside (0, 1) the term is nan -- followed, not repaired
    assert math.isnan(float(CL.sparse_loss(torch.tensor([-2.0, 0.5]), 0.1)))


class _Cfg:
    use_pearson_depth = True
    local_patch_size = 16
    depth_loss_stop_iteration = 100
    use_scaled_est_depth = True
    use_depth_regularization = False
    using_tv_loss = True


def test_optional_depth_terms_follow_the_models_switches_and_step_limits():
    pred, gt, img = (torch.from_numpy(a) for a in _images(48, 64, seed=4))
    terms = CL.optional_depth_terms(_Cfg, 50, pred[..., None], gt, img, torch.Generator().manual_seed(1), (1.0, 0.0))
    assert set(terms) == {"depth_local_pearson", "log_depth", "tv_loss"}
    late = CL.optional_depth_terms(_Cfg, 100, pred[..., None], gt, img, None, (1.0, 0.0))
    assert set(late) == {"log_depth", "tv_loss"}                    # Pearson stops at depth_loss_stop_iteration (:479)
    assert set(CL.optional_depth_terms(_Cfg, 20_000, pred, gt, img, None, None)) == set()  # no scale in the batch; TV < 20 000

    class Reg(_Cfg):
        use_depth_regularization = True

    with pytest.raises(NotImplementedError, match="Canny"):
        CL.optional_depth_terms(Reg, 50, pred, gt, img, None, None)


def test_cogs_loop_trains_with_the_optional_terms_switched_on(monkeypatch):
    import cpu_standins as SI
    import harness.pipeline as HP
    import harness.train as HT
    from oracle import oracle as O

    O.set_threads(4)
    monkeypatch.setattr(HP, "project_gaussians", SI.project_gaussians)
    monkeypatch.setattr(HP, "spherical_harmonics", SI.spherical_harmonics)
    monkeypatch.setattr(HP, "rasterize_gaussians", SI.rasterize_gaussians)
    seen = {"scale": 0, "terms": []}
    real_scale, real_terms = HT.cogs_losses.scale_regularisation, HT.cogs_losses.optional_depth_terms

    def scale(log_scales, ratio):
        seen["scale"] += 1
        return real_scale(log_scales, ratio)

    def terms(cfg, step, *a, **k):
        out = real_terms(cfg, step, *a, **k)
        seen["terms"].append((step, tuple(sorted(out))))
        return out

    monkeypatch.setattr(HT.cogs_losses, "scale_regularisation", scale)
    monkeypatch.setattr(HT.cogs_losses, "optional_depth_terms", terms)
    cfg = HT.TrainConfig(model="co-gs", num_gaussians=300, width=64, height=48, num_views=3, iters=24, sh_degree=1,
                         sh_degree_interval=10, eval_views=3, scene_scale=(0.03, 0.15), depth_loss_start_iteration=9,
                         background_color="random", densify=False, use_scale_regularization=True, use_est_depth=True,
                         use_pearson_depth=False, local_patch_size=16, use_scaled_est_depth=True, using_tv_loss=True)
    # (the Pearson term stays off in this loop: a patch of constant depth -- the background of a 300-Gaussian toy scene
    #  -- is 0 / 0 in the source's formula, and the restatement follows it; see the test above)
    res = HT.train(cfg, torch.device("cpu"))
    assert seen["scale"] == 3                                              # steps 0, 10, 20 (:450)
    assert [s for s, _ in seen["terms"]] == list(range(10, 24))            # step > depth_loss_start_iteration (:472-476)
    assert {t for _, t in seen["terms"]} == {("log_depth", "tv_loss")}
    assert np.isfinite(res["param_checksum"])
