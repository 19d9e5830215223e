"""GPU: BASELINE config 5's model, co-gs (`DepthGSModel`, gs_toolkit/models/depth_gs.py) -- the depth image rasterised
on the training path, the photometric loss as the source computes it (0.8 L1, the SSIM dropped, :445-448) and the
depth L1 (:531-538) -- one step against the ORACLE chain, the L1 head against torch ops, and short training runs."""
import numpy as np
import pytest
import torch

from harness import scene as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,clamp", [((37, 53, 3), False), ((64, 80, 3), True), ((5, 7, 3), True),
                                          ((1080, 1920, 3), True)])
def test_l1_head_equals_torch_ops(shape, clamp):
    from gs_fused import l1_loss

    g = torch.Generator(device=DEV).manual_seed(shape[0])
    pred = (torch.rand(shape, device=DEV, generator=g) * 1.3).requires_grad_(True)
    gt = torch.rand(shape, device=DEV, generator=g)
    ref_in = pred.detach().clone().requires_grad_(True)
    x = torch.clamp(ref_in, max=1.0) if clamp else ref_in
    want = 0.8 * torch.abs(gt - x).mean()
    want.backward()
    mine = l1_loss(pred, gt, 0.8, clamp_pred=clamp)
    (3.0 * mine).backward()
    assert abs(float(mine) - float(want)) < 2e-6 * max(1.0, abs(float(want)))
    assert torch.allclose(pred.grad, 3.0 * ref_in.grad, rtol=1e-6, atol=1e-12)
    if clamp:
        assert bool((pred.grad[pred.detach() > 1.0] == 0).all()) and bool((pred.detach() > 1.0).any())


def test_l1_head_rejects_bad_arguments():
    from gs_fused import l1_loss

    with pytest.raises(ValueError):
        l1_loss(torch.zeros(4, 4, 3, device=DEV), torch.zeros(4, 5, 3, device=DEV))
    with pytest.raises(RuntimeError):
        l1_loss(torch.zeros(4, 4, 3), torch.zeros(4, 4, 3))


def _cogs_cotangents(rgb, alpha, depth_acc, target, gt_depth, w_l1):
    """d (w_l1 |target - min(rgb, 1)|.mean() + |gt (gt > 0) - pred (gt > 0)|.mean()) / d (rgb, alpha, depth_acc) with
    pred = depth_acc / alpha where alpha > 0 (the far value elsewhere is detached), in float64 numpy."""
    H, W = alpha.shape
    x = np.minimum(rgb.astype(np.float64), 1.0)
    v_img = w_l1 * np.sign(x - target) / (3.0 * H * W)
    v_img[rgb > 1.0] = 0.0
    a = alpha.astype(np.float64)
    live = (a > 0) & (gt_depth > 0)
    inv = np.where(live, 1.0 / np.where(a > 0, a, 1.0), 0.0)
    pred = depth_acc.astype(np.float64) * inv
    vp = np.where(live, np.sign(pred - gt_depth), 0.0) / (H * W)
    v_dep = vp * inv
    v_alpha = -vp * pred * inv
    far = depth_acc.max()
    pred_full = np.where(a > 0, depth_acc / np.where(a > 0, a, 1.0), far)
    loss = w_l1 * np.abs(x - target).mean() + (np.abs(gt_depth - pred_full) * (gt_depth > 0)).mean()
    return v_img.astype(np.float32), v_alpha.astype(np.float32), v_dep.astype(np.float32), loss


@pytest.mark.parametrize("fused", [True, False])
def test_one_cogs_training_step_against_the_oracle_chain(fused):
    """RGB + depth through the separate ops as the co-gs caller issues them (fused: one RGB+depth compositing pass and
    the two fused loss heads; unfused: `rasterize_gaussians` twice and torch ops), both losses, backward -- against
    the oracle's forward images and the oracle's backward of the cotangents the two losses produce.  Tolerances:
    images 1e-4 abs on decision-stable pixels, loss 1e-5 relative, gradients 1e-3 relative (floor 1e-3 of the largest)."""
    from gs_fused import activate_gaussians, depth_l1_loss, l1_loss
    from harness.pipeline import CameraTensors, render_view
    from harness.train import blob_scene, cogs_depth_l1, orbit_cameras, quantise_depth_mm
    from test_gpu_render import _oracle_view

    W = H = 256
    n, deg_use, w_l1 = 10_000, 2, 0.8
    raw = blob_scene(n, seed=21, sh_degree=3)
    cam_np = orbit_cameras(8, W, H)[2]
    cam = CameraTensors.from_numpy(cam_np, DEV)
    bg = np.array(S.BACKGROUND, np.float32)
    rng = np.random.default_rng(9)
    target = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    zeros = np.zeros((H, W), np.float32)
    fwd = _oracle_view(raw, cam_np, bg, deg_use, H, W, np.zeros((H, W, 3), np.float32), zeros, zeros)
    # a "sensor" depth: the oracle's own normalised depth, perturbed, with holes (0 = no measurement)
    a0 = fwd["alpha"]
    gt_depth = np.where(a0 > 0.3, fwd["depth"] / np.maximum(a0, 1e-6) + rng.normal(0, 0.2, (H, W)), 0.0).astype(np.float32)
    gt_depth[rng.uniform(size=(H, W)) < 0.1] = 0.0
    gt_depth = quantise_depth_mm(torch.from_numpy(gt_depth)).numpy()
    v_img, v_alpha, v_dep, loss_ref = _cogs_cotangents(fwd["rgb"], fwd["alpha"], fwd["depth"], target.astype(np.float64),
                                                       gt_depth.astype(np.float64), w_l1)
    ref = _oracle_view(raw, cam_np, bg, deg_use, H, W, v_img, v_alpha, v_dep)

    p = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in raw.items()}
    scales, quats, opac, dirs = activate_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], cam.campos)
    out = render_view(p["means"], scales, quats, opac, (p["features_dc"], p["features_rest"]), cam,
                      torch.from_numpy(bg).to(DEV), deg_use, render_depth=True, retain_xys_grad=True, viewdirs=dirs,
                      clamp_rgb=not fused, fused_depth=fused, normalise_depth=not fused)
    t_target, t_gtd = torch.from_numpy(target).to(DEV), torch.from_numpy(gt_depth).to(DEV)
    if fused:
        loss = l1_loss(out["rgb"], t_target, w_l1, clamp_pred=True) + depth_l1_loss(out["depth_acc"], out["alpha"], t_gtd)
    else:
        loss = w_l1 * torch.abs(t_target - out["rgb"]).mean() + cogs_depth_l1(out["depth"], t_gtd)
    loss.backward()
    torch.cuda.synchronize()
    npy = lambda t: t.detach().cpu().numpy()
    ok = ref["ok"]
    assert ok.mean() > 0.98
    rgb = npy(out["rgb"])
    assert np.abs(np.minimum(rgb, 1.0) - np.minimum(ref["rgb"], 1.0))[ok].max() < 1e-4
    assert np.abs(npy(out["alpha"])[..., 0] - ref["alpha"])[ok].max() < 1e-4
    assert np.abs(npy(out["depth_acc"])[..., 0] - ref["depth"])[ok].max() < 1e-4 * max(1.0, float(ref["depths"].max()))
    assert abs(float(loss) - loss_ref) < 1e-5 * max(1.0, abs(loss_ref)), (float(loss), loss_ref)
    for k, g_ref in ref["grads"].items():
        mine = npy(p[k].grad).reshape(g_ref.shape)
        floor = 1e-3 * max(1e-9, float(np.abs(g_ref).max()))
        e = np.abs(mine - g_ref) / np.maximum(np.abs(g_ref), floor)
        # (unfused: the cotangents that enter the rasterizer's backward come out of the CALLER's fp32 torch chain --
        # where / div / abs / mean and their autograd -- while the checker forms them in float64: 2e-3 there)
        assert e.max() < (1e-3 if fused else 2e-3), f"{k}: max rel err {e.max():.3e}"
    # the depth loss really reaches the geometry: without it the gradient of the means is a different one
    no_dep = _oracle_view(raw, cam_np, bg, deg_use, H, W, v_img, zeros, None)
    assert np.abs(ref["grads"]["means"] - no_dep["grads"]["means"]).max() > 1e-2 * np.abs(ref["grads"]["means"]).max()


@pytest.mark.parametrize("fused", [True, False])
def test_short_cogs_run_learns_colour_and_depth(fused):
    """60 k Gaussians, 400 iterations, the depth loss from iteration 101, refinement active (compressed schedule):
    PSNR and the rendered depth's error against the ground-truth depth both improve; with the depth loss switched
    off the depth error is worse."""
    from gs_fused import RefineConfig
    from harness.train import TrainConfig, train

    rcfg = RefineConfig(warmup_length=60, refine_every=40, reset_alpha_every=30, stop_screen_size_at=300,
                        stop_split_at=360)
    kw = dict(model="co-gs", num_gaussians=60_000, init_gaussians=30_000, width=320, height=180, num_views=8,
              iters=400 if fused else 200, sh_degree=3, sh_degree_interval=80, densify=True, refine=rcfg,
              background_color="random", depth_loss_start_iteration=100 if fused else 50, log_every=10,
              fused_depth=fused, fused_loss=fused)
    res = train(TrainConfig(**kw), torch.device("cuda", 0))
    assert res["model"] == "co-gs" and res["depth"]["one_compositing_pass"] == fused
    assert np.isfinite(res["param_checksum"]) and res["peak_memory_bytes"] > 0
    e0, e1 = res["depth"]["mean_abs_error_start_end"]
    assert e1 < 0.6 * e0, (e0, e1)
    assert res["psnr_end"] > res["psnr_start"] + 2.0, res
    assert len(res["refinements"]) >= 3 and res["list_overflow_views"] == 0
    if fused:
        off = train(TrainConfig(**dict(kw, use_depth_loss=False)), torch.device("cuda", 0))
        assert off["depth"]["mean_abs_error_start_end"][1] > 1.5 * e1, (off["depth"], e1)
