"""CPU: the co-gs (`DepthGSModel`) training mode of the harness -- BASELINE config 5's loop -- on the oracle-backed
stand-ins of the native ops (tests/cpu_standins.py).  What is pinned here is the host logic the reference's source
prescribes (gs_toolkit/models/depth_gs.py): depth rasterised on the training path, the photometric loss WITHOUT its
SSIM term (:447-448), the depth L1 added unweighted from `step > depth_loss_start_iteration` (:472-476, :532-538).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.fixture()
def standins(monkeypatch):
    import cpu_standins as SI
    import harness.pipeline as HP
    import harness.train as HT
    from oracle import oracle as O

    O.set_threads(4)
    monkeypatch.setattr(HP, "project_gaussians", SI.project_gaussians)
    monkeypatch.setattr(HP, "spherical_harmonics", SI.spherical_harmonics)
    monkeypatch.setattr(HP, "rasterize_gaussians", SI.rasterize_gaussians)
    monkeypatch.setattr(HT, "_refine", lambda params, moments, stats, rcfg, step, ntd, max_dim, seed:
                        SI.refine_gaussians(params, moments, stats, rcfg, step, ntd, max_dim, seed=seed))
    return HT


def test_main_loss_drops_the_ssim_term_as_the_source_does():
    """depth_gs.py:445-448: `loss_dict["main_loss"] = (1 - ssim_lambda) * Ll1` and, on the next line,
    `+self.config.ssim_lambda * simloss` -- an expression statement.  The loss is 0.8 L1, whatever the SSIM is."""
    import harness.train as HT

    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(40, 56, 3, generator=g), torch.rand(40, 56, 3, generator=g)
    want = 0.8 * (a - b).abs().mean()
    assert torch.equal(HT.cogs_main_loss(a, b, 0.2), want + torch.tensor(0.0))
    with_ssim = 0.8 * (a - b).abs().mean() + 0.2 * (1 - HT.ssim(b, a))
    assert float(with_ssim) > float(want) + 0.05  # the term that is NOT there would have been large


def test_depth_l1_is_masked_by_the_ground_truth_and_averaged_over_all_pixels():
    import harness.train as HT

    g = torch.Generator().manual_seed(4)
    pred = torch.rand(12, 10, 1, generator=g) * 5
    gt = torch.rand(12, 10, generator=g) * 5
    gt[gt < 2.0] = 0.0  # no measurement
    m = gt > 0
    want = (gt - pred[..., 0]).abs()[m].sum() / gt.numel()  # mean over ALL pixels, zeros where gt == 0
    assert abs(float(HT.cogs_depth_l1(pred, gt)) - float(want)) < 1e-6


def test_quantised_ground_truth_depth_is_in_millimetres():
    import harness.train as HT

    d = torch.tensor([0.0, 1.23456, 70.0, -1.0])
    assert torch.allclose(HT.quantise_depth_mm(d), torch.tensor([0.0, 1.235, 65.535, 0.0]))


def test_cogs_training_loop_adds_the_depth_loss_after_its_start_iteration(standins):
    HT = standins
    seen = {"depth_calls": [], "main_calls": 0}
    real_depth, real_main = HT.cogs_depth_l1, HT.cogs_main_loss

    def depth(pred, gt):
        seen["depth_calls"].append(tuple(pred.shape))
        return real_depth(pred, gt)

    def main(pred, target, lam):
        seen["main_calls"] += 1
        return real_main(pred, target, lam)

    HT.cogs_depth_l1, HT.cogs_main_loss = depth, main
    try:
        cfg = HT.TrainConfig(model="co-gs", num_gaussians=500, width=64, height=48, num_views=4, iters=40,
                             sh_degree=1, sh_degree_interval=10, eval_views=4,
                             scene_scale=(0.03, 0.15), depth_loss_start_iteration=14, background_color="random",
                             log_every=1)
        res = HT.train(cfg, torch.device("cpu"))
    finally:
        HT.cogs_depth_l1, HT.cogs_main_loss = real_depth, real_main
    assert res["model"] == "co-gs" and res["depth"]["loss_from_step"] == 15
    assert seen["main_calls"] == 40
    assert len(seen["depth_calls"]) == 40 - 15 and set(seen["depth_calls"]) == {(48, 64, 1)}  # steps 15 .. 39
    # the depth term (an error in scene units, weight 1) jumps in on top of the photometric one (0.8 L1 < 0.1)
    assert res["losses"][15] > 1.3 * res["losses"][14], res["losses"][12:18]
    e0, e1 = res["depth"]["mean_abs_error_start_end"]
    assert np.isfinite(e0) and np.isfinite(e1) and e1 < 0.5 * e0, (e0, e1)
    assert np.isfinite(res["param_checksum"]) and res["psnr_end"] > res["psnr_start"]


def test_cogs_rejects_the_one_op_render_path(standins):
    HT = standins
    with pytest.raises(ValueError, match="co-gs"):
        HT.train(HT.TrainConfig(model="co-gs", num_gaussians=100, width=32, height=32, num_views=2, iters=1,
                                fused_render=True), torch.device("cpu"))
    with pytest.raises(ValueError, match="unknown model"):
        HT.train(HT.TrainConfig(model="nerfacto", num_gaussians=100, width=32, height=32, num_views=2, iters=1),
                 torch.device("cpu"))
