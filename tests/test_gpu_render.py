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

    raw = blob_scene(n, seed=seed, sh_degree={4: 1, 9: 2, 16: 3}[K])
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


@pytest.mark.parametrize("K,deg,render_depth", [(16, 3, False), (16, 1, True), (4, 1, False), (9, 2, True)])
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
        assert all(torch.isfinite(pc[k].grad).all() for k in pc) and pc["means"].grad.abs().sum().item() > 0
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
