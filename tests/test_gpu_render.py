"""gs_fused.render_gaussians (one autograd node for a whole view) and ViewGraph (one HIP
graph per view: render -> loss -> backward): same values as the separate ops."""
import numpy as np
import pytest
import torch

from harness import scene as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(n, K, seed):
    from harness.train import blob_scene

    raw = blob_scene(n, seed=seed, sh_degree={1: 0, 4: 1, 9: 2, 16: 3}[K])
    return {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in raw.items()}


def _camera(W, H, i=3):
    from harness.pipeline import CameraTensors
    from harness.train import orbit_cameras

    return CameraTensors.from_numpy(orbit_cameras(8, W, H)[i], DEV)


def _separate_ops(p, cam, bg, deg, render_depth):
    """The call sequence of the toolkit's models through the separate ops."""
    from gs_fused import activate_gaussians
    from harness.pipeline import render_view

    scales, quats, opac, dirs = activate_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], cam.campos)
    return render_view(p["means"], scales, quats, opac, (p["features_dc"], p["features_rest"]), cam, bg, deg,
                       render_depth=render_depth, retain_xys_grad=True, viewdirs=dirs, clamp_rgb=False,
                       fused_depth=True)


@pytest.mark.parametrize("K,deg,render_depth", [(16, 3, False), (16, 1, True), (4, 1, False), (9, 2, True),
                                                  (1, 0, False), (1, 0, True)])  # K = 1: BASELINE config 1 (SH degree 0)
def test_render_gaussians_equals_the_separate_ops(K, deg, render_depth):
    from gs_fused import DensifyStats, ViewSpec, densify_stats_, render_gaussians

    W, H, n = 320, 176, 30_000
    cam = _camera(W, H)
    bg = torch.tensor(S.BACKGROUND, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    v_img = torch.rand(H, W, 3, device=DEV, generator=g) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV, generator=g) * 2 - 1
    v_dep = torch.rand(H, W, device=DEV, generator=g) * 2 - 1

    pa = _model(n, K, seed=4)
    ref = _separate_ops(pa, cam, bg, deg, render_depth)
    outs, cots = [ref["rgb"], ref["alpha"][..., 0]], [v_img, v_alpha]
    if render_depth:
        # render_view divides by alpha; the fused op returns the accumulated depth: compare that
        pass
    torch.autograd.backward(outs, cots)
    stats_ref = DensifyStats(n, DEV, max(W, H))
    densify_stats_(ref["xys"].grad, ref["radii"], max(W, H), *stats_ref.as_tuple(), first=True)

    pb = _model(n, K, seed=4)
    spec = ViewSpec(H, W, cam.fx, cam.fy, cam.cx, cam.cy, deg, render_depth=render_depth)
    stats = DensifyStats(n, DEV, max(W, H))
    out = render_gaussians(pb["means"], pb["scales"], pb["quats"], pb["opacities"], pb["features_dc"],
                           pb["features_rest"], cam.viewmat, cam.projmat, cam.campos, bg, spec, capacity=4_000_000,
                           stats=stats)
    torch.cuda.synchronize()
    count = int(out["count"][0])
    assert 0 < count < 4_000_000
    assert torch.equal(out["rgb"], ref["rgb"]) and torch.equal(out["alpha"], ref["alpha"][..., 0])
    assert torch.equal(out["radii"], ref["radii"])
    torch.autograd.backward([out["rgb"], out["alpha"]], [v_img, v_alpha])
    for k in pa:
        a, b = pa[k].grad, pb[k].grad
        if a.numel() == 0:  # features_rest of an SH-degree-0 model is [N, 0, 3]
            assert b.shape == a.shape
            continue
        assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item() + 1e-12, k
    for a, b in zip(stats_ref.as_tuple(), stats.as_tuple()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-9)
    assert int(stats.first[0]) == 0
    if render_depth:
        # the depth image and its gradient against the fused RGBD op used directly
        pc = _model(n, K, seed=4)
        out2 = render_gaussians(pc["means"], pc["scales"], pc["quats"], pc["opacities"], pc["features_dc"],
                                pc["features_rest"], cam.viewmat, cam.projmat, cam.campos, bg, spec, capacity=count)
        d_ref = ref["depth"][..., 0] * ref["alpha"][..., 0]  # undo the division where alpha > 0
        ok = ref["alpha"][..., 0] > 0
        assert (out2["depth"] - d_ref)[ok].abs().max().item() < 1e-4 * float(out2["depth"].max())
        out2["depth"].backward(v_dep)
        # value check: the same cotangent through the separate ops' accumulated depth (depth * alpha undoes
        # render_view's division; its gradient flows through both factors)
        pd = _model(n, K, seed=4)
        ref2 = _separate_ops(pd, cam, bg, deg, render_depth)
        (ref2["depth"][..., 0] * ref2["alpha"][..., 0]).backward(v_dep * ok)
        pe = _model(n, K, seed=4)
        out4 = render_gaussians(pe["means"], pe["scales"], pe["quats"], pe["opacities"], pe["features_dc"],
                                pe["features_rest"], cam.viewmat, cam.projmat, cam.campos, bg, spec, capacity=count)
        out4["depth"].backward(v_dep * ok)
        for k in ("means", "scales", "quats", "opacities"):
            a, b = pd[k].grad, pe[k].grad
            assert (a - b).abs().max().item() <= 1e-3 * a.abs().max().item() + 1e-12, k
        assert pe["means"].grad.abs().sum().item() > 0
        # a capacity that is too small is reported, not hidden
        out3 = render_gaussians(pc["means"], pc["scales"], pc["quats"], pc["opacities"], pc["features_dc"],
                                pc["features_rest"], cam.viewmat, cam.projmat, cam.campos, bg, spec,
                                capacity=count // 2)
        torch.cuda.synchronize()
        assert int(out3["count"][0]) == count > count // 2


def test_view_graph_replays_equal_eager_views():
    """render -> L1+SSIM -> backward captured as one HIP graph; replays over different cameras
    and targets reproduce the eager results (images bit-identical, gradients to atomics order)."""
    from gs_fused import DensifyStats, ViewSpec, l1_ssim_loss, render_gaussians
    from gs_fused.render import ViewGraph

    W, H, n, K, deg = 256, 144, 20_000, 16, 2
    bg = torch.tensor(S.BACKGROUND, device=DEV)
    cams = [_camera(W, H, i) for i in range(4)]
    spec = ViewSpec(H, W, cams[0].fx, cams[0].fy, cams[0].cx, cams[0].cy, deg)
    g = torch.Generator(device=DEV).manual_seed(1)
    gts = [torch.rand(H, W, 3, device=DEV, generator=g) for _ in cams]
    loss_fn = lambda out, targets: l1_ssim_loss(out["rgb"], targets[0], 0.2, clamp_pred=True)

    p = _model(n, K, seed=6)
    stats = DensifyStats(n, DEV, max(W, H))
    vg = ViewGraph(p, spec, 3_000_000, loss_fn, bg, [(H, W, 3)], stats=stats)
    vg.capture(cams[0].viewmat, cams[0].projmat, cams[0].campos, (gts[0],))
    q = _model(n, K, seed=6)
    stats_e = DensifyStats(n, DEV, max(W, H))
    for i, (cam, gt) in enumerate(zip(cams, gts)):
        loss, out = vg.replay(cam.viewmat, cam.projmat, cam.campos, (gt,))
        for t in q.values():
            t.grad = None
        oe = render_gaussians(q["means"], q["scales"], q["quats"], q["opacities"], q["features_dc"],
                              q["features_rest"], cam.viewmat, cam.projmat, cam.campos, bg, spec, 3_000_000,
                              stats=stats_e)
        le = loss_fn(oe, (gt,))
        le.backward()
        torch.cuda.synchronize()
        assert vg.fits()
        assert torch.equal(out["rgb"], oe["rgb"]) and abs(float(loss) - float(le)) < 1e-6
        for k in p:
            a, b = q[k].grad, p[k].grad
            assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item() + 1e-12, (i, k)
    for a, b in zip(stats_e.as_tuple(), stats.as_tuple()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-9)
    # a graph captured with too small a capacity says so after the replay
    small = ViewGraph(p, spec, 1000, loss_fn, bg, [(H, W, 3)])
    small.capture(cams[0].viewmat, cams[0].projmat, cams[0].campos, (gts[0],))
    small.replay(cams[1].viewmat, cams[1].projmat, cams[1].campos, (gts[1],))
    torch.cuda.synchronize()
    assert not small.fits()


def test_render_gaussians_with_random_list_capacities():
    """tools/exp/fuzz_render.py: capacities that are too small (down to one entry), exact and ample, on random
    view sizes, SH degrees and depth on / off: the reported count never depends on the capacity, fitting
    capacities give the same image and gradients, cut lists stay finite and memory-safe through the backward."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_render.py"), "30", "71"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") == 30


def _oracle_view(raw, cam_np, bg, deg_use, H, W, v_img, v_alpha, v_dep=None):
    """The whole view through the ORACLE (oracle/gsr_oracle.c + the caller's arithmetic of
    vanilla_gs.py:765-857 in numpy): images, the six parameter gradients, densification statistics."""
    from oracle import oracle as O
    from oracle.refine import update_stats

    f32 = np.float32
    n = raw["means"].shape[0]
    K = raw["features_rest"].shape[1] + 1
    degree = {4: 1, 9: 2, 16: 3}[K]
    scales = np.exp(raw["scales"]).astype(f32)
    qn = np.linalg.norm(raw["quats"].astype(np.float64), axis=-1, keepdims=True)
    quats = (raw["quats"] / qn).astype(f32)
    opac = (1.0 / (1.0 + np.exp(-raw["opacities"].astype(np.float64)))).astype(f32)
    d = raw["means"] - cam_np.campos[None]
    dirs = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(f32)
    coeffs = np.concatenate([raw["features_dc"][:, None, :], raw["features_rest"]], axis=1)
    sh = O.compute_sh_forward(n, degree, deg_use, dirs, coeffs)
    rgbs = np.maximum(sh + 0.5, 0).astype(f32)
    r = O.render_forward(raw["means"], scales, 1.0, quats, cam_np.viewmat[:3], cam_np.projmat, cam_np.fx, cam_np.fy,
                         cam_np.cx, cam_np.cy, H, W, 16, rgbs, opac, bg, ambig_eps=1e-5)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    a = O.rasterize_backward(H, W, 16, r["gaussian_ids_sorted"], r["tile_bins"], r["xys"], r["conics"], rgbs, opac, bg,
                             r["final_Ts"], r["final_idx"], v_img, v_alpha)
    vxy, vconic, vcol, vop = a
    v_depth = np.zeros(n, f32)
    out = {"rgb": r["out_img"], "alpha": 1 - r["final_Ts"], "ok": ~r["ambig"], "radii": r["radii"], "depths": r["depths"]}
    if v_dep is not None:
        zero3 = np.zeros(3, f32)
        dcol = np.repeat(r["depths"][:, None], 3, 1).astype(f32)
        dimg, dT, dI = O.rasterize_forward(tb, (16, 16, 1), (W, H, 1), r["gaussian_ids_sorted"], r["tile_bins"], r["xys"],
                                           r["conics"], dcol, opac, zero3)
        vd3 = np.zeros((H, W, 3), f32)
        vd3[..., 0] = v_dep
        b = O.rasterize_backward(H, W, 16, r["gaussian_ids_sorted"], r["tile_bins"], r["xys"], r["conics"], dcol, opac,
                                 zero3, dT, dI, vd3, np.zeros((H, W), f32))
        vxy, vconic, vop = vxy + b[0], vconic + b[1], vop + b[3]
        v_depth = b[2][:, 0].astype(f32)
        out["depth"] = dimg[..., 0]
    v_sh = (vcol * (sh + 0.5 > 0)).astype(f32)
    v_coeffs = O.compute_sh_backward(n, degree, deg_use, dirs, v_sh)
    _, _, v_mean, v_scale, v_quat = O.project_gaussians_backward(
        n, raw["means"], scales, 1.0, quats, cam_np.viewmat[:3], cam_np.projmat, cam_np.fx, cam_np.fy, cam_np.cx,
        cam_np.cy, H, W, r["cov3d"], r["radii"], r["conics"], r["compensation"], vxy.astype(f32), v_depth,
        vconic.astype(f32), np.zeros(n, f32))
    vq = v_quat.astype(np.float64)
    q64 = quats.astype(np.float64)
    o64 = opac.astype(np.float64)
    out["grads"] = {
        "means": v_mean, "scales": v_scale * scales,
        "quats": ((vq - (vq * q64).sum(-1, keepdims=True) * q64) / qn).astype(f32),
        "opacities": (vop.reshape(n, 1) * o64 * (1 - o64)).astype(f32),
        "features_dc": v_coeffs[:, 0, :], "features_rest": v_coeffs[:, 1:, :],
    }
    out["stats"] = update_stats(None, vxy, r["radii"], max(W, H))
    return out


@pytest.mark.parametrize("render_depth", [False, True])
def test_render_gaussians_against_the_oracle_chain(render_depth):
    """`render_gaussians` (one autograd node for the whole view) against the ORACLE, not against this
    package's separate ops: RGB, alpha, accumulated depth, all six parameter gradients -- the depth
    cotangent included -- and the densification statistics, on a config-1-sized scene (10 k Gaussians,
    256 x 256).  Tolerances: images 1e-4 abs on decision-stable pixels, gradients 1e-3 relative
    (floor 1e-3 of the largest), statistics 1e-3 relative."""
    from gs_fused import DensifyStats, ViewSpec, render_gaussians
    from harness.train import blob_scene, orbit_cameras
    from harness.pipeline import CameraTensors

    W = H = 256
    n, deg_use = 10_000, 2
    raw = blob_scene(n, seed=11, sh_degree=3)
    cam_np = orbit_cameras(8, W, H)[5]
    cam = CameraTensors.from_numpy(cam_np, DEV)
    bg = np.array(S.BACKGROUND, np.float32)
    rng = np.random.default_rng(5)
    v_img = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
    v_alpha = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    v_dep = rng.uniform(-1, 1, (H, W)).astype(np.float32) if render_depth else None
    ref = _oracle_view(raw, cam_np, bg, deg_use, H, W, v_img, v_alpha, v_dep)

    p = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in raw.items()}
    spec = ViewSpec(H, W, cam.fx, cam.fy, cam.cx, cam.cy, deg_use, render_depth=render_depth)
    stats = DensifyStats(n, DEV, max(W, H))
    out = render_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], p["features_dc"], p["features_rest"],
                           cam.viewmat, cam.projmat, cam.campos, torch.from_numpy(bg).to(DEV), spec,
                           capacity=2_000_000, stats=stats)
    npy = lambda t: t.detach().cpu().numpy()
    ok = ref["ok"]
    assert ok.mean() > 0.98
    assert np.array_equal(npy(out["radii"]), ref["radii"])
    assert np.abs(npy(out["rgb"]) - ref["rgb"])[ok].max() < 1e-4
    assert np.abs(npy(out["alpha"]) - ref["alpha"])[ok].max() < 1e-4
    outs, cots = [out["rgb"], out["alpha"]], [torch.from_numpy(v_img).to(DEV), torch.from_numpy(v_alpha).to(DEV)]
    if render_depth:
        assert np.abs(npy(out["depth"]) - ref["depth"])[ok].max() < 1e-4 * max(1.0, float(ref["depths"].max()))
        outs.append(out["depth"])
        cots.append(torch.from_numpy(v_dep).to(DEV))
    torch.autograd.backward(outs, cots)
    torch.cuda.synchronize()
    for k, g_ref in ref["grads"].items():
        mine = npy(p[k].grad).reshape(g_ref.shape)
        floor = 1e-3 * max(1e-9, float(np.abs(g_ref).max()))
        e = np.abs(mine - g_ref) / np.maximum(np.abs(g_ref), floor)
        assert e.max() < 1e-3, f"{k}: max rel err {e.max():.3e}"
    if render_depth:  # the depth cotangent reaches the means (a finiteness check says nothing about that)
        no_dep = _oracle_view(raw, cam_np, bg, deg_use, H, W, v_img, v_alpha, None)
        assert np.abs(ref["grads"]["means"] - no_dep["grads"]["means"]).max() > 1e-2 * np.abs(ref["grads"]["means"]).max()
    for mine, want in zip(stats.as_tuple(), ref["stats"]):
        assert np.allclose(npy(mine).astype(np.float32), want, rtol=1e-3, atol=1e-3 * float(np.abs(want).max()) + 1e-12)


def test_view_graph_replay_against_the_oracle():
    """One HIP-graph replay of a view (render -> linear loss -> backward) against the ORACLE chain:
    captured on one camera, replayed on another -- image and all six parameter gradients."""
    from gs_fused import ViewSpec
    from gs_fused.render import ViewGraph
    from harness.pipeline import CameraTensors
    from harness.train import blob_scene, orbit_cameras

    W, H, n, deg_use = 192, 128, 6_000, 3
    raw = blob_scene(n, seed=13, sh_degree=3)
    cams_np = orbit_cameras(8, W, H)
    bg = np.array(S.BACKGROUND, np.float32)
    rng = np.random.default_rng(8)
    v_img = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
    v_alpha = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    p = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in raw.items()}
    c0, c1 = (CameraTensors.from_numpy(cams_np[i], DEV) for i in (1, 6))
    spec = ViewSpec(H, W, c0.fx, c0.fy, c0.cx, c0.cy, deg_use)
    loss_fn = lambda out, targets: (out["rgb"] * targets[0]).sum() + (out["alpha"] * targets[1]).sum()
    vg = ViewGraph(p, spec, 1_500_000, loss_fn, torch.from_numpy(bg).to(DEV), [(H, W, 3), (H, W)])
    t_img, t_alpha = torch.from_numpy(v_img).to(DEV), torch.from_numpy(v_alpha).to(DEV)
    vg.capture(c0.viewmat, c0.projmat, c0.campos, (t_img, t_alpha))
    loss, out = vg.replay(c1.viewmat, c1.projmat, c1.campos, (t_img, t_alpha))
    torch.cuda.synchronize()
    assert vg.fits()
    ref = _oracle_view(raw, cams_np[6], bg, deg_use, H, W, v_img, v_alpha, None)
    ok = ref["ok"]
    assert np.abs(out["rgb"].detach().cpu().numpy() - ref["rgb"])[ok].max() < 1e-4
    for k, g_ref in ref["grads"].items():
        mine = p[k].grad.detach().cpu().numpy().reshape(g_ref.shape)
        floor = 1e-3 * max(1e-9, float(np.abs(g_ref).max()))
        e = np.abs(mine - g_ref) / np.maximum(np.abs(g_ref), floor)
        assert e.max() < 1e-3, f"{k}: max rel err {e.max():.3e}"
