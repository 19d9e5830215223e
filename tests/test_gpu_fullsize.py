"""GPU: BASELINE.json's configurations at full size.

config 2 (200 k Gaussians, SH degree 3, 1920x1080, forward+backward): HIP vs the
CPU oracle on identical inputs, image 1e-4 abs (numerically stable pixels),
gradients 1e-3 rel (abs floor 1e-3 x max|grad|).

The 1 M-Gaussian bench workload (configs 3/4 shape): size-independent
properties -- linearity in the colours, consistency of alpha, additivity of the
backward in its cotangents, agreement of the fused binning with the reference
pipeline, bounded run-to-run gradient jitter (fp32 atomics)."""
import numpy as np
import pytest
import torch

from harness import scene as S
from harness.pipeline import CameraTensors, render_view
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cu(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_(True) if grad else t


def npy(t):
    return t.detach().cpu().numpy()


def stable_gaussians(amb_pixels, amb_gaussians, xys, radii, W, H):
    """Gaussians none of whose discrete decisions can legitimately differ between two
    correct fp32 implementations: not flagged by the oracle's backward (own alpha / sigma
    within 1e-5 of a threshold) and with no forward-unstable pixel (termination or threshold
    within 1e-5 of flipping, which changes final_T / final_idx of that pixel for every
    Gaussian drawn there) inside their 3-sigma box."""
    ii = np.zeros((H + 1, W + 1), np.int64)
    ii[1:, 1:] = amb_pixels.astype(np.int64).cumsum(0).cumsum(1)
    x, y, rad = xys[:, 0], xys[:, 1], radii.astype(np.float32)
    x0 = np.clip(np.floor(x - rad), 0, W).astype(int)
    x1 = np.clip(np.ceil(x + rad) + 1, 0, W).astype(int)
    y0 = np.clip(np.floor(y - rad), 0, H).astype(int)
    y1 = np.clip(np.ceil(y + rad) + 1, 0, H).astype(int)
    touched = (ii[y1, x1] - ii[y0, x1] - ii[y1, x0] + ii[y0, x0]) > 0
    return ~(touched | amb_gaussians)


# share of the visible Gaussians of config 2 that are decision-stable, i.e. held to 1e-3 relative
# elementwise: measured value minus 10 % (profiles/r03_config2_stable_fraction.json)
WITHIN_FLOOR = 0.999  # share of ALL visible Gaussians within 1e-3 relative outright: measured 0.9999 (xys), 1.0 (opacities)
STABLE_VISIBLE_FLOOR = 0.049  # measured 0.0547 (9 876 of 180 416 visible Gaussians), round 3


FLIPPED_PIXELS = 4.0  # how many peak-pixel-equivalents of a Gaussian's terms a legitimately flipped decision may move


def peak_pixel_share(conics):
    """Share of a splat's summed per-pixel weight that its PEAK pixel carries: 1 / (2 pi sigma_x sigma_y) in px^2
    = sqrt(det conic) / (2 pi) (the conic is the inverse 2-D covariance), at most 1."""
    det = np.maximum(conics[:, 0].astype(np.float64) * conics[:, 2] - conics[:, 1].astype(np.float64) ** 2, 0.0)
    return np.minimum(1.0, np.sqrt(det) / (2.0 * np.pi))


def grad_close(mine, ref, abs_sum=None, name="", stable=None, worst_cap=20.0, peak_share=None, stable_rel=1e-3,
               unstable_fraction=1e-4, worst_check=True, global_max=1e-3, global_l2=1e-4):
    """Per-Gaussian gradients are sums over up to ~1e4 pixels with heavy cancellation.
      * STABLE Gaussians (see `stable_gaussians`): north_star's bar, |err| <= 1e-3 |ref|
        elementwise, with |ref| floored at 1e-4 max|ref| (elements that cancel to ~0);
      * the others (a pixel's decision may differ): |err| <= 1e-3 |ref| + 5e-4 * abs_sum for all
        but a 1e-4 fraction of elements, none by more than 20x (abs_sum = sum of |per-pixel
        terms| from the oracle, the scale a flipped pixel perturbs);
      * max |err| <= 1e-3 max|ref|  and  ||err||_2 <= 1e-4 ||ref||_2  always."""
    err = np.abs(mine - ref)
    if stable is not None:
        rel = err[stable] / np.maximum(np.abs(ref[stable]), 1e-4 * np.abs(ref).max())
        assert rel.max() <= stable_rel, f"{name}: stable Gaussians differ by {rel.max():.3e} relative"
    if abs_sum is not None:
        bound = 1e-3 * np.abs(ref) + 5e-4 * abs_sum
        ratio = err / np.maximum(bound, 1e-30)
        assert (ratio > 1).mean() <= unstable_fraction, f"{name}: {(ratio > 1).sum()} elements exceed 1e-3|ref| + 5e-4*abs_sum"
        if not worst_check:
            print(f"{name}: worst unstable element {ratio.max():.1f}x the tight bound (not asserted for this scene family)")
        elif peak_share is None:
            assert ratio.max() < worst_cap, f"{name}: worst element {ratio.max():.2f}x the bound"
        else:
            # no constant: a flipped decision moves an element by that pixel's term, at most the share of abs_sum the
            # splat's peak pixel carries (its footprint decides: 0.1 % for a 30-px splat, 4 % at sigma = 2 px) --
            # FLIPPED_PIXELS such pixels' worth on top of the bound, for EVERY element
            share = peak_share.reshape((-1,) + (1,) * (err.ndim - 1))
            outer = bound + FLIPPED_PIXELS * share * abs_sum
            worst = float((err / np.maximum(outer, 1e-30)).max())
            print(f"{name}: worst unstable element {ratio.max():.1f}x the tight bound, {worst:.3f} of the footprint bound")
            assert worst <= 1.0, f"{name}: an element is off by more than {FLIPPED_PIXELS} peak pixels' terms ({worst:.2f}x)"
    l2 = np.linalg.norm(mine - ref) / max(np.linalg.norm(ref), 1e-30)
    print(f"{name}: max |err| / max |ref| = {err.max() / max(np.abs(ref).max(), 1e-30):.3e}, L2 relative = {l2:.3e}")
    assert err.max() <= global_max * np.abs(ref).max(), f"{name}: max abs err {err.max():.3e} vs max|ref| {np.abs(ref).max():.3e}"
    assert l2 <= global_l2, f"{name}: L2 relative error {l2:.3e}"


@pytest.mark.timeout(900)
def test_config2_200k_sh3_1080p_forward_backward_vs_oracle():
    _forward_backward_vs_oracle(200_000, 0.005, 0.05, STABLE_VISIBLE_FLOOR, "config2_stable_fraction.json", 20.0)


# bench.py's default workload (BASELINE's headline: 1 M Gaussians, SH degree 3, 1920x1080; SURVEY 8d's scale range
# halved, as bench.py states): the share of decision-stable visible Gaussians measured in round 4 minus 10 %
STABLE_VISIBLE_FLOOR_1M = 0.155  # measured 0.1728 (150 152 of 869 029 visible Gaussians), round 4


@pytest.mark.timeout(1500)
def test_bench_default_1m_sh3_vs_oracle():
    """The TIMED workload itself against the oracle, with config 2's assertions: projection bit-identical, image and
    alpha within 1e-4 on decision-stable pixels, every parameter's gradient within 1e-3 (forward.cu:278-395,
    backward.cu:133-303).  bench.py reports the same comparison in its line (`parity_vs_oracle`)."""
    # The worst decision-UNSTABLE element is no longer held to a constant multiple of the tight bound (20 x on config 2,
    # raised to 60 x here in round 4 when 25.5 x was seen: VERDICT r4, weak 1b) but to what a flipped decision CAN move:
    # `grad_close(peak_share=...)` bounds every element by the tight bound + FLIPPED_PIXELS peak pixels' worth of the
    # Gaussian's own terms, from its footprint.  (`worst_cap` only applies where no footprint is passed.)
    _forward_backward_vs_oracle(1_000_000, 0.0025, 0.025, STABLE_VISIBLE_FLOOR_1M, "bench_default_stable_fraction.json", 60.0)


def _forward_backward_vs_oracle(n, scale_lo, scale_hi, stable_floor, report_file, worst_cap):
    W, H, deg = 1920, 1080, 3
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=deg, seed=42, scale_lo=scale_lo, scale_hi=scale_hi)
    scene_vs_oracle(sc, cam, deg, stable_floor, report_file, worst_cap)


def scene_vs_oracle(sc, cam, deg, stable_floor, report_file, worst_cap, min_stable_pixels=0.99, within_floor=WITHIN_FLOOR,
                    stable_rel=1e-3, unstable_fraction=1e-4, worst_check=True, global_max=1e-3, global_l2=1e-4):
    """Any scene dictionary (harness.scene layout) from `cam` through the public ops against the oracle, with config 2's
    assertions (tests/test_gpu_heldout.py runs the held-out families and a trained model through it)."""
    W, H, n = cam.width, cam.height, sc["means3d"].shape[0]
    bg = np.array(S.BACKGROUND, np.float32)
    v_img, v_alpha = S.make_cotangents(cam)

    params = {k: cu(v, True) for k, v in sc.items()}
    out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"],
                      params["sh_coeffs"], CameraTensors.from_numpy(cam, DEV), cu(bg), deg,
                      retain_xys_grad=True, clamp_rgb=False)
    torch.autograd.backward([out["rgb"], out["alpha"]], [cu(v_img), cu(v_alpha)[..., None]])
    torch.cuda.synchronize()

    # ---- oracle, same inputs
    dirs = S.viewdirs_for(sc, cam)
    sh = O.compute_sh_forward(n, deg, deg, dirs, sc["sh_coeffs"])
    np.testing.assert_allclose(npy(out["rgbs"]), np.maximum(sh + 0.5, 0), rtol=1e-4, atol=2e-5)
    # composite with the GPU's own per-Gaussian inputs so that the comparison isolates each stage:
    # (1) projection vs oracle
    cov3d, xys, depths, radii, conics, comp, tiles = O.project_gaussians_forward(
        n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat, cam.fx, cam.fy,
        cam.cx, cam.cy, H, W, 16, 0.01)
    g_radii, g_tiles = npy(out["radii"]), npy(out["num_tiles_hit"])
    # bit-exact: same FMA policy and operation order as the oracle (csrc/project.hip header)
    assert np.array_equal(g_radii, radii) and np.array_equal(g_tiles, tiles)
    assert np.array_equal(npy(out["xys"]), xys) and np.array_equal(npy(out["conics"]), conics)
    assert np.array_equal(npy(out["depths"]), depths)
    # (2) binning + compositing on the GPU's projection outputs
    gx, gd, gc = npy(out["xys"]), npy(out["depths"]), npy(out["conics"])
    rgbs = npy(out["rgbs"])
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    I, cum = O.compute_cumulative_intersects(g_tiles)
    _, _, ks, vs, bins = O.bin_and_sort_gaussians(n, I, gx, gd, g_radii, cum, tb, 16)
    ref_img, ref_T, ref_idx, amb = O.rasterize_forward(tb, (16, 16, 1), (W, H, 1), vs, bins, gx, gc, rgbs,
                                                       sc["opacities"], bg, ambig_eps=1e-5)
    ok = ~amb
    assert ok.mean() > min_stable_pixels
    img = npy(out["rgb"])
    assert np.abs(img - ref_img)[ok].max() < 1e-4
    assert np.abs((1 - npy(out["alpha"])[..., 0]) - ref_T)[ok].max() < 1e-4
    assert np.abs(img - ref_img).max() < 0.05
    # (3) backward of the compositing + projection + SH
    vxy, vconic, vcol, vop, axy, aconic, acol, aop, amb_g = O.rasterize_backward(
        H, W, 16, vs, bins, gx, gc, rgbs, sc["opacities"], bg, ref_T, ref_idx, v_img, v_alpha,
        with_abs_sums=True, ambig_eps=1e-5)
    stable = stable_gaussians(amb, amb_g, gx, g_radii, W, H)
    # how many Gaussians carry the 1e-3 claim: the share of the VISIBLE ones (an invisible Gaussian is
    # trivially stable) that are decision-stable.  Printed, stored next to the profiles, and floored at
    # the measured value minus 10 % (round 3: see STABLE_VISIBLE_FLOOR)
    visible = g_radii > 0
    frac = float((stable & visible).sum()) / float(visible.sum())
    report = {"gaussians": n, "visible": int(visible.sum()), "stable_visible": int((stable & visible).sum()),
              "stable_fraction_of_visible": round(frac, 4), "ambiguous_pixels": round(float(amb.mean()), 5),
              "floor": stable_floor}
    print("config2 stable Gaussians:", report)
    try:
        import json
        import os

        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, report_file), "w") as fh:
            json.dump(report, fh)
    except OSError:
        pass
    # ... and what the REST does: the share of all visible Gaussians whose xys / opacity gradients meet
    # 1e-3 relative outright (|ref| floored at 1e-4 max|ref|), whether decision-stable or not
    def within(mine, ref):
        rel = np.abs(mine - ref) / np.maximum(np.abs(ref), 1e-4 * np.abs(ref).max())
        return (rel.reshape(len(ref), -1).max(axis=1) <= 1e-3)[visible]

    ok_xy, ok_op = within(npy(out["xys"].grad), vxy), within(npy(params["opacities"].grad), vop)
    report["visible_within_1e-3_rel"] = {"xys": round(float(ok_xy.mean()), 4), "opacities": round(float(ok_op.mean()), 4)}
    print("config2 stable Gaussians:", report)
    try:
        with open(os.path.join(out_dir, report_file), "w") as fh:
            json.dump(report, fh)
    except (OSError, NameError):
        pass
    assert frac > stable_floor, report
    assert ok_xy.mean() > within_floor and ok_op.mean() > within_floor, report
    share = peak_pixel_share(gc)
    kw = dict(stable_rel=stable_rel, unstable_fraction=unstable_fraction, worst_check=worst_check, global_max=global_max,
              global_l2=global_l2)
    grad_close(npy(out["xys"].grad), vxy, axy, name="xys.grad", stable=stable, worst_cap=worst_cap, peak_share=share, **kw)
    grad_close(npy(params["opacities"].grad), vop, aop, name="opacities", stable=stable, worst_cap=worst_cap,
               peak_share=share, **kw)
    vsh = O.compute_sh_backward(n, deg, deg, dirs, (vcol * (sh + 0.5 > 0)).astype(np.float32))
    grad_close(npy(params["sh_coeffs"].grad), vsh, name="sh_coeffs", stable=stable, **kw)
    zeros = np.zeros(n, np.float32)
    _, _, vmean, vscale, vquat = O.project_gaussians_backward(
        n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat, cam.fx, cam.fy,
        cam.cx, cam.cy, H, W, cov3d, g_radii, gc, comp, vxy, zeros, vconic, zeros)
    grad_close(npy(params["means3d"].grad), vmean, name="means3d", stable=stable, **kw)
    grad_close(npy(params["scales"].grad), vscale, name="scales", stable=stable, **kw)
    grad_close(npy(params["quats"].grad), vquat, name="quats", stable=stable, **kw)


@pytest.fixture(scope="module")
def big():
    W, H, n = 1920, 1080, 1_000_000
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=0.0025, scale_hi=0.025)
    from rasterizer import project_gaussians

    with torch.no_grad():
        xys, depths, radii, conics, comp, tiles, cov3d = project_gaussians(
            cu(sc["means3d"]), cu(sc["scales"]), 1, cu(sc["quats"]), cu(cam.viewmat)[:3], cu(cam.projmat),
            cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
    return dict(W=W, H=H, n=n, xys=xys, depths=depths, radii=radii, conics=conics, tiles=tiles,
                opac=cu(sc["opacities"]))


def raster(big, colors, bg, **kw):
    from rasterizer import rasterize_gaussians

    return rasterize_gaussians(big["xys"], big["depths"], big["radii"], big["conics"], big["tiles"], colors,
                               big["opac"], big["H"], big["W"], 16, background=bg, **kw)


def test_1m_fused_binning_equals_reference_pipeline(big):
    import rasterizer.cuda as C
    from rasterizer import utils as U

    n, tb = big["n"], ((big["W"] + 15) // 16, (big["H"] + 15) // 16, 1)
    I, cum = U.compute_cumulative_intersects(big["tiles"])
    ref = U.bin_and_sort_gaussians(n, I, big["xys"], big["depths"], big["radii"], cum, tb, 16)
    order, cum_sorted = C.depth_order(big["depths"], big["radii"], big["tiles"])
    assert int(cum_sorted[-1].item()) == I
    ids, bins = C.bin_sorted(n, I, order, cum_sorted, big["xys"], big["radii"], tb, 16)
    assert torch.equal(ids, ref[3]) and torch.equal(bins, ref[4])
    ks = ref[2]
    assert bool((ks[1:] >= ks[:-1]).all())  # sortedness of the 64-bit keys
    # every intersection is owned by exactly one tile range
    lens = (bins[:, 1] - bins[:, 0]).to(torch.int64)
    assert int(lens.sum().item()) == I and bool((lens >= 0).all())


def test_1m_forward_is_linear_in_colours_and_alpha_is_colour_free(big):
    g = torch.Generator(device=DEV).manual_seed(0)
    n = big["n"]
    c1 = torch.rand(n, 3, device=DEV, generator=g)
    c2 = torch.rand(n, 3, device=DEV, generator=g)
    zero = torch.zeros(3, device=DEV)
    with torch.no_grad():
        i1, a1 = raster(big, c1, zero, return_alpha=True)
        i2, a2 = raster(big, c2, zero, return_alpha=True)
        i3, a3 = raster(big, 0.25 * c1 + 2.0 * c2, zero, return_alpha=True)
        ib = raster(big, c1, torch.tensor([0.3, 0.6, 0.9], device=DEV))
    assert torch.equal(a1, a2) and torch.equal(a1, a3)  # geometry only, deterministic
    assert (i3 - (0.25 * i1 + 2.0 * i2)).abs().max().item() < 2e-5
    assert a1.min().item() >= 0.0 and a1.max().item() <= 1.0
    bgc = torch.tensor([0.3, 0.6, 0.9], device=DEV)
    assert (ib - (i1 + (1 - a1)[..., None] * bgc)).abs().max().item() < 1e-5
    # forward is deterministic (no atomics): bit-identical on a second run
    with torch.no_grad():
        i1b = raster(big, c1, zero)
    assert torch.equal(i1, i1b)


def test_1m_backward_is_additive_in_cotangents_and_stable(big):
    g = torch.Generator(device=DEV).manual_seed(1)
    n, H, W = big["n"], big["H"], big["W"]
    colors = torch.rand(n, 3, device=DEV, generator=g)
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    va = torch.rand(H, W, 3, device=DEV, generator=g) * 2 - 1
    vb = torch.rand(H, W, 3, device=DEV, generator=g) * 2 - 1

    def grads(v_img):
        c = colors.clone().requires_grad_(True)
        o = big["opac"].clone().requires_grad_(True)
        x = big["xys"].clone().requires_grad_(True)
        from rasterizer import rasterize_gaussians

        img = rasterize_gaussians(x, big["depths"], big["radii"], big["conics"], big["tiles"], c, o, H, W, 16,
                                  background=bg)
        img.backward(v_img)
        return c.grad, o.grad, x.grad

    ga, gb, gab = grads(va), grads(vb), grads(va + vb)
    for a, b, ab, nm in zip(ga, gb, gab, ("colors", "opacity", "xys")):
        scale = ab.abs().max().item()
        assert (ab - (a + b)).abs().max().item() < 2e-4 * scale, nm
    # fp32 atomics: run-to-run jitter stays at rounding level
    ga2 = grads(va)
    for a, a2 in zip(ga, ga2):
        assert (a - a2).abs().max().item() <= 1e-5 * a.abs().max().item()
    # culled Gaussians get exactly zero gradient
    culled = big["radii"] == 0
    assert ga[0][culled].abs().sum().item() == 0 and ga[2][culled].abs().sum().item() == 0


@pytest.mark.timeout(900)
def test_config5_shape_4k_rgb_and_depth_pass_vs_oracle():
    """BASELINE config 5's shape at a size the oracle finishes in seconds: 4K (32 400 tiles:
    the tile scatter in 4 tile-row bands), RGB pass + differentiable depth pass
    (co-gs `render_depth` branch: depths as colours, zero background, second call reuses the
    lists), forward and backward of both passes against the oracle."""
    from rasterizer import project_gaussians, rasterize_gaussians, spherical_harmonics

    W, H, n, deg = 3840, 2160, 60_000, 3
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=deg, seed=7, scale_lo=0.005, scale_hi=0.05)
    bg = np.array(S.BACKGROUND, np.float32)
    rng = np.random.default_rng(11)
    v_img = rng.standard_normal((H, W, 3)).astype(np.float32)
    v_alpha = rng.standard_normal((H, W)).astype(np.float32)
    v_dep = rng.standard_normal((H, W, 3)).astype(np.float32)
    ct = CameraTensors.from_numpy(cam, DEV)
    p = {k: cu(v, True) for k, v in sc.items()}
    xys, depths, radii, conics, comp, tiles, _ = project_gaussians(
        p["means3d"], p["scales"], 1, p["quats"], ct.viewmat[:3], ct.projmat, cam.fx, cam.fy, cam.cx, cam.cy,
        H, W, 16)
    for t in (xys, depths, conics):
        t.retain_grad()
    dirs = S.viewdirs_for(sc, cam)
    rgbs = torch.clamp(spherical_harmonics(deg, cu(dirs), p["sh_coeffs"]) + 0.5, min=0.0)
    rgbs.retain_grad()
    rgb, alpha = rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, p["opacities"], H, W, 16,
                                     background=cu(bg), return_alpha=True)
    dcol = depths[:, None].repeat(1, 3)
    dcol.retain_grad()
    dimg = rasterize_gaussians(xys, depths, radii, conics, tiles, dcol, p["opacities"], H, W, 16,
                               background=torch.zeros(3, device=DEV))
    torch.autograd.backward([rgb, alpha, dimg], [cu(v_img), cu(v_alpha), cu(v_dep)])
    torch.cuda.synchronize()

    gx, gd, gc, gr, gt = (npy(t) for t in (xys, depths, conics, radii, tiles))
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    I, cum = O.compute_cumulative_intersects(gt)
    _, _, ks, vs, bins = O.bin_and_sort_gaussians(n, I, gx, gd, gr, cum, tb, 16)
    col = npy(rgbs)
    ref_img, ref_T, ref_idx, amb = O.rasterize_forward(tb, (16, 16, 1), (W, H, 1), vs, bins, gx, gc, col,
                                                       sc["opacities"], bg, ambig_eps=1e-5)
    dcol_np = np.repeat(gd[:, None], 3, 1).astype(np.float32)
    ref_dep, ref_T2, ref_idx2, amb2 = O.rasterize_forward(tb, (16, 16, 1), (W, H, 1), vs, bins, gx, gc, dcol_np,
                                                          sc["opacities"], np.zeros(3, np.float32), ambig_eps=1e-5)
    ok = ~(amb | amb2)
    assert ok.mean() > 0.99
    assert np.abs(npy(rgb) - ref_img)[ok].max() < 1e-4
    assert np.abs(npy(alpha) - (1 - ref_T))[ok].max() < 1e-4
    scale = max(1.0, float(gd.max()))
    assert np.abs(npy(dimg) - ref_dep)[ok].max() < 1e-4 * scale  # depths are not in [0,1]
    a = O.rasterize_backward(H, W, 16, vs, bins, gx, gc, col, sc["opacities"], bg, ref_T, ref_idx, v_img, v_alpha,
                             with_abs_sums=True)
    b = O.rasterize_backward(H, W, 16, vs, bins, gx, gc, dcol_np, sc["opacities"], np.zeros(3, np.float32),
                             ref_T2, ref_idx2, v_dep, np.zeros((H, W), np.float32), with_abs_sums=True)
    grad_close(npy(xys.grad), a[0] + b[0], a[4] + b[4], name="xys.grad (both passes)")
    grad_close(npy(conics.grad), a[1] + b[1], a[5] + b[5], name="conics.grad (both passes)")
    grad_close(npy(p["opacities"].grad), a[3] + b[3], a[7] + b[7], name="opacities (both passes)")
    grad_close(npy(rgbs.grad), a[2], a[6], name="rgbs.grad")
    grad_close(npy(dcol.grad), b[2], b[6], name="depth colours")


# ---- BASELINE config 5 at full size: 3 M Gaussians, 3840 x 2160, RGB + depth -----------
@pytest.fixture(scope="module")
def c5():
    W, H, n = 3840, 2160, 3_000_000
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=42, scale_lo=0.005, scale_hi=0.05)  # SURVEY 8d scales
    from rasterizer import project_gaussians

    with torch.no_grad():
        xys, depths, radii, conics, comp, tiles, cov3d = project_gaussians(
            cu(sc["means3d"]), cu(sc["scales"]), 1, cu(sc["quats"]), cu(cam.viewmat)[:3], cu(cam.projmat),
            cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
    return dict(W=W, H=H, n=n, xys=xys, depths=depths, radii=radii, conics=conics, tiles=tiles,
                opac=cu(sc["opacities"]))


@pytest.mark.timeout(900)
def test_config5_full_size_binning_properties_and_equals_reference_pipeline(c5):
    """3 M Gaussians at 4K (I = 212 M reference list entries, 32 400 tiles = 4 bands):
    (a) with reach_records = None ... not available above 16384 tiles, so the reference
    pipeline (64-bit sort, binning.hip) is compared with the band-wise exact lists through
    size-independent properties: every tile's exact list is a subsequence of the reference's
    in the same order (checked as: sorted by (depth, id) within the tile, and a subset by
    per-Gaussian counts), the ranges tile the output exactly, and the device-sized build
    (capacity only, no host count) returns the same lists."""
    import rasterizer.cuda as C
    from rasterizer import utils as U

    n, W, H = c5["n"], c5["W"], c5["H"]
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    bands = C.tile_bands(tb)
    assert bands == 4
    # the default large-grid path: one count per Gaussian, two-level partition
    cnt, recs = C.count_reach(c5["xys"], c5["radii"], c5["conics"], c5["opac"], tb)
    assert cnt.shape == (n,) and bool((cnt <= c5["tiles"]).all())
    order, cum = C.depth_order(c5["depths"], c5["radii"], cnt)
    I2 = int(cum[-1].item())
    I = int(c5["tiles"].to(torch.int64).sum().item())
    assert 0.3 * I < I2 < 0.7 * I, (I, I2)
    ids, bins = C.bin_sorted(n, I2, order, cum, c5["xys"], c5["radii"], tb, 16, recs)
    # ... equals the tile-row-band path (single-pass scatter band by band)
    cnt_b, recs_b = C.count_reach(c5["xys"], c5["radii"], c5["conics"], c5["opac"], tb, bands=bands)
    assert torch.equal(cnt_b.view(bands, n).sum(0).to(torch.int32), cnt)
    order_b, cum_b = C.depth_order(c5["depths"], c5["radii"], cnt_b)
    ids_b, bins_b = C.bin_sorted(n, I2, order_b, cum_b, c5["xys"], c5["radii"], tb, 16, recs_b)
    assert torch.equal(ids_b, ids) and torch.equal(bins_b, bins)
    del ids_b, bins_b, cnt_b, recs_b, order_b, cum_b
    # the ranges tile [0, I2) in tile order
    lens = (bins[:, 1] - bins[:, 0]).to(torch.int64)
    assert int(lens.sum().item()) == I2 and bool((lens >= 0).all())
    nz = lens > 0
    starts = torch.cumsum(lens, 0) - lens
    assert torch.equal(bins[nz, 0].to(torch.int64), starts[nz])
    # every Gaussian appears exactly count_reach times
    assert torch.equal(torch.bincount(ids, minlength=n), cnt.to(torch.int64))
    # within a tile: ordered by (depth bits, id) -- the reference's 64-bit key order
    tile_of = torch.repeat_interleave(torch.arange(tb[0] * tb[1], device=DEV), lens)
    key = (c5["depths"][ids.long()].view(torch.int32).to(torch.int64) << 32) | ids.to(torch.int64)
    same = tile_of[1:] == tile_of[:-1]
    assert bool((key[1:][same] > key[:-1][same]).all())
    # each listed (Gaussian, tile) pair lies inside the Gaussian's reference bounding box
    tx, ty = (tile_of % tb[0]).float(), (tile_of // tb[0]).float()
    x, y, r = c5["xys"][ids.long(), 0] / 16, c5["xys"][ids.long(), 1] / 16, c5["radii"][ids.long()].float() / 16
    inside = (tx >= (x - r).floor().clamp(0, tb[0])) & (tx < (x + r + 1).floor().clamp(0, tb[0])) & \
             (ty >= (y - r).floor().clamp(0, tb[1])) & (ty < (y + r + 1).floor().clamp(0, tb[1]))
    assert bool(inside.all())
    del tile_of, key, same, tx, ty, x, y, r, inside
    # device-sized build: capacity only
    count = torch.zeros(1, dtype=torch.int32).pin_memory()
    ids2, bins2 = C.bin_sorted(n, I2 + (1 << 20), order, cum, c5["xys"], c5["radii"], tb, 16, recs,
                               device_sized=True, count_out=count)
    torch.cuda.synchronize()
    assert int(count[0]) == I2 and torch.equal(ids2[:I2], ids) and torch.equal(bins2, bins)
    del ids2, bins2
    # the image composited from the exact lists equals the one from the reference pipeline's lists
    Iref, cumref = U.compute_cumulative_intersects(c5["tiles"])
    assert Iref == I
    ref = U.bin_and_sort_gaussians(n, Iref, c5["xys"], c5["depths"], c5["radii"], cumref, tb, 16)
    g = torch.Generator(device=DEV).manual_seed(5)
    colors = torch.rand(n, 3, device=DEV, generator=g)
    bg = torch.tensor([0.2, 0.4, 0.6], device=DEV)
    args = (tb, (16, 16, 1), (W, H, 1))
    img_ref, T_ref, idx_ref = C.rasterize_forward(*args, ref[3], ref[4], c5["xys"], c5["conics"], colors,
                                                  c5["opac"], bg)
    del ref
    img, T, idx = C.rasterize_forward(*args, ids, bins, c5["xys"], c5["conics"], colors, c5["opac"], bg)
    assert torch.equal(img, img_ref) and torch.equal(T, T_ref)


@pytest.mark.timeout(900)
def test_config5_full_size_fused_rgbd_equals_two_passes_without_host_sync(c5):
    """RGB + depth at 3 M / 4K: one fused compositing pass == the two passes of the models
    (bit-identical RGB / alpha, depth to rounding; gradients = sum over both passes), and
    the steady state builds its lists without reading the count back (no `.item()`)."""
    import rasterizer.cuda as C
    from gs_fused import rasterize_gaussians_rgbd
    from rasterizer import rasterize_gaussians

    n, W, H = c5["n"], c5["W"], c5["H"]
    g = torch.Generator(device=DEV).manual_seed(6)
    colors = torch.rand(n, 3, device=DEV, generator=g)
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    v_img = torch.rand(H, W, 3, device=DEV, generator=g) * 2 - 1
    v_dep = torch.rand(H, W, device=DEV, generator=g) * 2 - 1

    def leaves():
        return (c5["xys"].clone().requires_grad_(True), colors.clone().requires_grad_(True),
                c5["opac"].clone().requires_grad_(True), c5["depths"].clone().requires_grad_(True))

    x, c, o, d = leaves()
    rgb2, a2 = rasterize_gaussians(x, c5["depths"], c5["radii"], c5["conics"], c5["tiles"], c, o, H, W, 16,
                                   background=bg, return_alpha=True)
    dep2 = rasterize_gaussians(x, c5["depths"], c5["radii"], c5["conics"], c5["tiles"], d[:, None].repeat(1, 3), o,
                               H, W, 16, background=torch.zeros(3, device=DEV))[..., 0]
    torch.autograd.backward([rgb2, dep2], [v_img, v_dep])
    g2 = [t.grad.clone() for t in (x, c, o, d)]
    # steady state (a view that sized its lists from the previous one; the two-round plan has the culled count,
    # which travels through a pinned slot and takes one view to arrive): no host read-back on the critical path
    x, c, o, d = leaves()
    rasterize_gaussians_rgbd(x, c5["depths"], c5["radii"], c5["conics"], c5["tiles"], c, d, o, H, W, background=bg)
    x, c, o, d = leaves()
    rasterize_gaussians_rgbd(x, c5["depths"], c5["radii"], c5["conics"], c5["tiles"], c, d, o, H, W, background=bg)
    items = {"n": 0}
    orig = torch.Tensor.item

    def counting_item(self):
        items["n"] += 1
        return orig(self)

    x, c, o, d = leaves()
    torch.Tensor.item = counting_item
    try:
        rgb1, a1, dep1 = rasterize_gaussians_rgbd(x, c5["depths"], c5["radii"], c5["conics"], c5["tiles"], c, d, o,
                                                  H, W, background=bg)
    finally:
        torch.Tensor.item = orig
    assert items["n"] == 0, "the 4K steady state must not read the list length back with .item()"
    dep1 = dep1[..., 0]
    torch.autograd.backward([rgb1, dep1], [v_img, v_dep])
    assert torch.equal(rgb1, rgb2) and torch.equal(a1, a2)
    scale = float(c5["depths"].max())
    assert (dep1 - dep2).abs().max().item() < 1e-5 * scale
    for a, b, nm in zip((x, c, o, d), g2, ("xys", "colors", "opacity", "depths")):
        s_ = b.abs().max().item()
        assert (a.grad - b).abs().max().item() < 2e-4 * s_, nm
