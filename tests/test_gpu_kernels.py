"""GPU: every HIP kernel, called through the C ABI (via `rasterizer.cuda`),
against the CPU oracle on the same seeded inputs.

Tolerances: integer / index outputs bit-exact; fp32 images 1e-4 abs on pixels
whose discrete decisions are numerically stable (the oracle flags the others:
|alpha - 1/255|, |T(1-alpha) - 1e-4| or |sigma| within 1e-5 relative -- a
1-ulp difference in exp() legitimately flips those); gradients 1e-3 relative
with an absolute floor of 1e-3 x max|grad| (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from harness import scene as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def npy(t):
    return t.detach().cpu().numpy()


def rel_err(a, b, floor):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def grad_close(mine, ref, tol=1e-3, name=""):
    floor = 1e-3 * max(1e-6, float(np.abs(ref).max()))
    e = rel_err(mine, ref, floor)
    assert e.max() < tol, f"{name}: max rel err {e.max():.3e} at {np.unravel_index(e.argmax(), e.shape)}"


def make(n, W, H, deg=3, seed=1, cam_kw=None, **kw):
    cam = S.make_camera(W, H, **(cam_kw or {}))
    sc = S.make_scene(n, cam, sh_degree=deg, seed=seed, **kw)
    return cam, sc


def project_cpu(cam, sc, bw=16, clip=0.01, glob=1.0):
    n = sc["means3d"].shape[0]
    return O.project_gaussians_forward(n, sc["means3d"], sc["scales"], glob, sc["quats"],
                                       cam.viewmat[:3], cam.projmat, cam.fx, cam.fy, cam.cx, cam.cy,
                                       cam.height, cam.width, bw, clip)


def project_gpu(cam, sc, bw=16, clip=0.01, glob=1.0):
    import rasterizer.cuda as C

    n = sc["means3d"].shape[0]
    return C.project_gaussians_forward(n, cu(sc["means3d"]), cu(sc["scales"]), glob, cu(sc["quats"]),
                                       cu(cam.viewmat[:3]), cu(cam.projmat), cam.fx, cam.fy, cam.cx,
                                       cam.cy, cam.height, cam.width, bw, clip)


CASES = [
    # n, W, H, bw, camera kwargs
    (10_000, 256, 256, 16, {}),
    (3_000, 200, 120, 16, dict(yaw=0.2, pitch=-0.1, roll=0.05, trans=(0.3, -0.2, 0.5))),
    (2_000, 97, 61, 8, dict(yaw=-0.3)),
    (500, 33, 47, 5, {}),
]


@pytest.mark.parametrize("n,W,H,bw,ck", CASES)
def test_project_forward(n, W, H, bw, ck):
    cam, sc = make(n, W, H, cam_kw=ck, scale_lo=0.01, scale_hi=0.2)
    ref = project_cpu(cam, sc, bw)
    out = [npy(t) for t in project_gpu(cam, sc, bw)]
    names = ["cov3d", "xys", "depths", "radii", "conics", "compensation", "num_tiles_hit"]
    r = dict(zip(names, ref))
    o = dict(zip(names, out))
    # project.hip and the oracle are both compiled without FMA contraction and evaluate the
    # same expressions in the same order with correctly rounded / and sqrt: every output is
    # BIT-IDENTICAL, the integer ones (radii sit behind a ceil) included
    for k in names:
        assert np.array_equal(o[k], r[k]), f"{k}: {(o[k] != r[k]).sum()} elements differ"
    # culled splats: everything the consumers read is zero
    cul = r["radii"] == 0
    for k in ("xys", "depths", "compensation", "num_tiles_hit"):
        assert np.all(o[k][cul] == 0), k


@pytest.mark.parametrize("n,W,H,bw,ck", CASES)
def test_project_backward(n, W, H, bw, ck):
    import rasterizer.cuda as C

    cam, sc = make(n, W, H, cam_kw=ck, scale_lo=0.01, scale_hi=0.2)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    rng = np.random.default_rng(5)
    v_xy = rng.standard_normal((n, 2)).astype(np.float32)
    v_depth = rng.standard_normal(n).astype(np.float32)
    v_conic = rng.standard_normal((n, 3)).astype(np.float32)
    v_comp = rng.standard_normal(n).astype(np.float32)
    ref = O.project_gaussians_backward(n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3],
                                       cam.projmat, cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width,
                                       cov3d, radii, conics, comp, v_xy, v_depth, v_conic, v_comp)
    out = C.project_gaussians_backward(n, cu(sc["means3d"]), cu(sc["scales"]), 1.0, cu(sc["quats"]),
                                       cu(cam.viewmat[:3]), cu(cam.projmat), cam.fx, cam.fy, cam.cx, cam.cy,
                                       cam.height, cam.width, cu(cov3d), cu(radii), cu(conics), cu(comp),
                                       cu(v_xy), cu(v_depth), cu(v_conic), cu(v_comp))
    for o, r, nm in zip(out, ref, ["v_cov2d", "v_cov3d", "v_mean3d", "v_scale", "v_quat"]):
        o = npy(o)
        assert np.all(o[radii <= 0] == 0), nm
        # per-row relative error (gradient magnitudes span many decades)
        rowmax = np.abs(r).max(axis=-1, keepdims=True)
        e = np.abs(o - r) / np.maximum(rowmax, 1e-6 * np.abs(r).max())
        assert e.max() < 1e-3, f"{nm}: {e.max():.3e}"


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("n", [1, 63, 4097])
def test_sh(deg, n):
    import rasterizer.cuda as C

    rng = np.random.default_rng(deg * 100 + n)
    K = (deg + 1) ** 2
    dirs = rng.standard_normal((n, 3)).astype(np.float32) * 3.0  # un-normalised on purpose
    coeffs = rng.standard_normal((n, K, 3)).astype(np.float32)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    for use in range(deg + 1):
        ref = O.compute_sh_forward(n, deg, use, dirs, coeffs)
        out = npy(C.compute_sh_forward(n, deg, use, cu(dirs), cu(coeffs)))
        np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)
        refb = O.compute_sh_backward(n, deg, use, dirs, v)
        outb = npy(C.compute_sh_backward(n, deg, use, cu(dirs), cu(v)))
        np.testing.assert_allclose(outb, refb, rtol=1e-4, atol=1e-6)
        assert np.all(outb[:, (use + 1) ** 2:] == 0)


@pytest.mark.parametrize("n,W,H,bw,ck", CASES)
def test_binning_bit_exact(n, W, H, bw, ck):
    """scan, key emission, sort, bin edges: integer work, bit-exact."""
    import rasterizer.cuda as C
    from rasterizer import utils as U

    cam, sc = make(n, W, H, cam_kw=ck, scale_lo=0.01, scale_hi=0.2)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    I, cum = O.compute_cumulative_intersects(tiles)
    I_g, cum_g = U.compute_cumulative_intersects(cu(tiles))
    assert I_g == I and np.array_equal(npy(cum_g), cum)
    ref = O.bin_and_sort_gaussians(n, I, xys, depths, radii, cum, tb, bw)
    out = U.bin_and_sort_gaussians(n, I, cu(xys), cu(depths), cu(radii), cum_g, tb, bw)
    for o, r, nm in zip(out, ref, ["isect", "gids", "isect_sorted", "gids_sorted", "tile_bins"]):
        assert np.array_equal(npy(o), r), nm
    ks = npy(out[2])
    assert np.all(np.diff(ks) >= 0)
    assert out[2].dtype == torch.int64 and out[3].dtype == torch.int32 and out[4].dtype == torch.int32


@pytest.mark.parametrize("n,W,H,bw,ck", CASES + [(200_000, 640, 360, 16, {}), (50_000, 2560, 1440, 16, {}),
                                             (30_000, 3840, 2160, 16, {}), (20_000, 5120, 2880, 16, {}), (20_000, 48, 32, 16, {}),
                                             (300_000, 64, 64, 4, {})])
def test_fused_binning_equals_reference_pipeline(n, W, H, bw, ck):
    """depth_order + bin_sorted (what rasterize_gaussians runs) produce the same
    gaussian_ids_sorted / tile_bins as scan + map + 64-bit sort + bin edges."""
    import rasterizer.cuda as C

    cam, sc = make(n, W, H, cam_kw=ck, scale_lo=0.01, scale_hi=0.2)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    # force depth ties: equal keys must keep ascending Gaussian id
    depths = depths.copy()
    depths[radii > 0] = np.round(depths[radii > 0] * 4) / 4
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    I, cum = O.compute_cumulative_intersects(tiles)
    _, _, ks, vs, bins = O.bin_and_sort_gaussians(n, I, xys, depths, radii, cum, tb, bw)
    order, cum_sorted = C.depth_order(cu(depths), cu(radii), cu(tiles))
    assert int(cum_sorted[-1].item()) == I
    o = npy(order)
    assert np.array_equal(np.sort(o), np.arange(n))
    dkey = np.where(radii > 0, depths, 0).astype(np.float32)
    assert np.all(np.diff(dkey[o]) >= 0)
    ids, tile_bins = C.bin_sorted(n, I, order, cum_sorted, cu(xys), cu(radii), tb, bw)
    assert np.array_equal(npy(ids), vs)
    assert np.array_equal(npy(tile_bins), bins)


def test_sort_is_stable_on_ties():
    """Equal (tile, depth) keys keep emission order (ascending Gaussian id)."""
    import rasterizer.cuda as C

    rng = np.random.default_rng(0)
    I = 100_000
    tiles = rng.integers(0, 37, I).astype(np.int64)
    depth = rng.integers(1, 9, I).astype(np.int64)  # many duplicates
    keys = (tiles << 32) | depth
    vals = np.arange(I, dtype=np.int32)
    ks, vs = C.sort_intersects(cu(keys), cu(vals), 37)
    rk, rv = O.sort_intersects(keys, vals)
    assert np.array_equal(npy(ks), rk) and np.array_equal(npy(vs), rv)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(rv, vals[order])


def raster_inputs(n, W, H, bw, ck, channels=3, seed=1, **kw):
    cam, sc = make(n, W, H, cam_kw=ck, seed=seed, **kw)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    I, cum = O.compute_cumulative_intersects(tiles)
    _, _, ks, vs, bins = O.bin_and_sort_gaussians(n, I, xys, depths, radii, cum, tb, bw)
    rng = np.random.default_rng(seed + 7)
    colors = rng.uniform(0, 1, (n, channels)).astype(np.float32)
    bg = rng.uniform(0, 1, channels).astype(np.float32)
    return dict(cam=cam, tb=tb, bw=bw, I=I, vs=vs, bins=bins, xys=xys, conics=conics, colors=colors,
                opac=sc["opacities"], bg=bg, n=n, W=W, H=H)


RASTER_CASES = [
    (10_000, 256, 256, 16, {}, dict(scale_lo=0.005, scale_hi=0.05)),
    (4_000, 200, 120, 16, dict(yaw=0.2, pitch=-0.1), dict(scale_lo=0.02, scale_hi=0.3)),  # dense: T<=1e-4 path
    (20_000, 97, 61, 16, {}, dict(scale_lo=0.01, scale_hi=0.1)),  # ragged edges, long lists
    (2_000, 97, 61, 8, dict(yaw=-0.3), dict(scale_lo=0.02, scale_hi=0.2)),
    (500, 33, 47, 5, {}, dict(scale_lo=0.02, scale_hi=0.2)),
]


def check_image(out, Ts, ref, amb):
    ok = ~amb
    assert ok.mean() > 0.98
    np.testing.assert_allclose(out[ok], ref[0][ok], rtol=0, atol=1e-4)
    np.testing.assert_allclose(Ts[ok], ref[1][ok], rtol=0, atol=1e-4)
    # unstable pixels may differ by one splat's contribution, never by garbage
    assert np.abs(out - ref[0]).max() < 0.05


@pytest.mark.parametrize("n,W,H,bw,ck,kw", RASTER_CASES)
def test_rasterize_forward(n, W, H, bw, ck, kw):
    import rasterizer.cuda as C

    d = raster_inputs(n, W, H, bw, ck, **kw)
    ref = O.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), d["vs"], d["bins"], d["xys"], d["conics"],
                              d["colors"], d["opac"], d["bg"], ambig_eps=1e-5)
    out, Ts, idx = C.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), cu(d["vs"]), cu(d["bins"]),
                                       cu(d["xys"]), cu(d["conics"]), cu(d["colors"]), cu(d["opac"]),
                                       cu(d["bg"]))
    check_image(npy(out), npy(Ts), ref, ref[3])
    ok = ~ref[3]
    assert np.array_equal(npy(idx)[ok], ref[2][ok])  # index of the last contributing splat
    assert idx.dtype == torch.int32


@pytest.mark.parametrize("n,W,H,bw,ck,kw", RASTER_CASES)
def test_rasterize_backward(n, W, H, bw, ck, kw):
    import rasterizer.cuda as C

    d = raster_inputs(n, W, H, bw, ck, **kw)
    out, Ts, idx = O.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), d["vs"], d["bins"], d["xys"],
                                       d["conics"], d["colors"], d["opac"], d["bg"])
    rng = np.random.default_rng(11)
    v_img = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
    v_alpha = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    ref = O.rasterize_backward(H, W, bw, d["vs"], d["bins"], d["xys"], d["conics"], d["colors"], d["opac"],
                               d["bg"], Ts, idx, v_img, v_alpha)
    got = C.rasterize_backward(H, W, bw, cu(d["vs"]), cu(d["bins"]), cu(d["xys"]), cu(d["conics"]),
                               cu(d["colors"]), cu(d["opac"]), cu(d["bg"]), cu(Ts), cu(idx), cu(v_img),
                               cu(v_alpha))
    assert got[3].shape == (n, 1)
    for g, r, nm in zip(got, ref, ["v_xy", "v_conic", "v_colors", "v_opacity"]):
        grad_close(npy(g), r, name=nm)


@pytest.mark.parametrize("channels", [1, 4, 7, 32, 33, 96])  # above 32: one pass per 32 channels
def test_nd_rasterize(channels):
    import rasterizer.cuda as C

    n, W, H, bw = 3000, 120, 90, 16
    d = raster_inputs(n, W, H, bw, {}, channels=channels, scale_lo=0.02, scale_hi=0.2)
    ref = O.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), d["vs"], d["bins"], d["xys"], d["conics"],
                              d["colors"], d["opac"], d["bg"], ambig_eps=1e-5)
    out, Ts, idx = C.nd_rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), cu(d["vs"]), cu(d["bins"]),
                                          cu(d["xys"]), cu(d["conics"]), cu(d["colors"]), cu(d["opac"]),
                                          cu(d["bg"]))
    check_image(npy(out), npy(Ts), ref, ref[3])
    rng = np.random.default_rng(3)
    v_img = rng.uniform(-1, 1, (H, W, channels)).astype(np.float32)
    v_alpha = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    refb = O.rasterize_backward(H, W, bw, d["vs"], d["bins"], d["xys"], d["conics"], d["colors"], d["opac"],
                                d["bg"], ref[1], ref[2], v_img, v_alpha)
    got = C.nd_rasterize_backward(H, W, bw, cu(d["vs"]), cu(d["bins"]), cu(d["xys"]), cu(d["conics"]),
                                  cu(d["colors"]), cu(d["opac"]), cu(d["bg"]), cu(ref[1]), cu(ref[2]),
                                  cu(v_img), cu(v_alpha))
    for g, r, nm in zip(got, refb, ["v_xy", "v_conic", "v_colors", "v_opacity"]):
        grad_close(npy(g), r, name=nm)


def test_tile16_matches_generic_kernel():
    """The wave-per-tile kernels and the lane-per-pixel kernels implement the same rule."""
    import rasterizer.cuda as C

    n, W, H, bw = 8000, 160, 112, 16
    d = raster_inputs(n, W, H, bw, {}, scale_lo=0.01, scale_hi=0.15)
    a = (d["tb"], (bw, bw, 1), (W, H, 1), cu(d["vs"]), cu(d["bins"]), cu(d["xys"]), cu(d["conics"]),
         cu(d["colors"]), cu(d["opac"]), cu(d["bg"]))
    o1 = C.rasterize_forward(*a)
    o2 = C.nd_rasterize_forward(*a)
    assert (o1[0] - o2[0]).abs().max().item() < 1e-5
    assert (o1[2] == o2[2]).float().mean().item() > 0.9999
    v_img = torch.rand(H, W, 3, device=DEV) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV) * 2 - 1
    b = (H, W, bw) + a[3:] + (o1[1], o1[2], v_img, v_alpha)
    g1 = C.rasterize_backward(*b)
    g2 = C.nd_rasterize_backward(*b)
    for x, y in zip(g1, g2):
        grad_close(npy(x), npy(y), tol=1e-3)


def test_backward_alpha_saturation_follows_cuda_rule():
    """opacity*exp(-sigma) > 0.99: forward clamps at 0.999, backward at 0.99
    (forward.cu:360 vs backward.cu:232).  Unpinned by the torch oracle; the C
    oracle follows the CUDA source and the HIP kernel must follow it too."""
    import rasterizer.cuda as C

    W = H = 32
    bw = 16
    n = 6
    xys = np.array([[8.2, 8.1], [9.0, 7.5], [20.3, 20.2], [21.0, 19.0], [8.0, 24.0], [24.0, 8.0]], np.float32)
    conics = np.tile(np.array([[0.05, 0.0, 0.05]], np.float32), (n, 1))
    opac = np.array([[0.9999], [0.6], [0.9999], [0.9999], [0.995], [0.5]], np.float32)
    colors = np.random.default_rng(0).uniform(0, 1, (n, 3)).astype(np.float32)
    depths = np.linspace(1, 2, n).astype(np.float32)
    radii = np.full(n, 30, np.int32)
    tb = (2, 2, 1)
    tiles = np.full(n, 4, np.int32)
    I, cum = O.compute_cumulative_intersects(tiles)
    _, _, ks, vs, bins = O.bin_and_sort_gaussians(n, I, xys, depths, radii, cum, tb, bw)
    bg = np.array([0.2, 0.3, 0.4], np.float32)
    out, Ts, idx = O.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), vs, bins, xys, conics, colors, opac, bg)
    rng = np.random.default_rng(2)
    v_img = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
    v_alpha = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    ref = O.rasterize_backward(H, W, bw, vs, bins, xys, conics, colors, opac, bg, Ts, idx, v_img, v_alpha)
    o = C.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), cu(vs), cu(bins), cu(xys), cu(conics), cu(colors),
                            cu(opac), cu(bg))
    np.testing.assert_allclose(npy(o[0]), out, atol=1e-4)
    got = C.rasterize_backward(H, W, bw, cu(vs), cu(bins), cu(xys), cu(conics), cu(colors), cu(opac), cu(bg),
                               cu(Ts), cu(idx), cu(v_img), cu(v_alpha))
    for g, r, nm in zip(got, ref, ["v_xy", "v_conic", "v_colors", "v_opacity"]):
        grad_close(npy(g), r, name=nm)


def test_cov2d_bounds():
    import rasterizer.cuda as C

    rng = np.random.default_rng(4)
    n = 1000
    a = rng.uniform(0.3, 50, n)
    c = rng.uniform(0.3, 50, n)
    b = rng.uniform(-0.9, 0.9, n) * np.sqrt(a * c)
    cov = np.stack([a, b, c], -1).astype(np.float32)
    ref = O.compute_cov2d_bounds(n, cov)
    got = C.compute_cov2d_bounds(n, cu(cov))
    np.testing.assert_allclose(npy(got[0]), ref[0], rtol=1e-4, atol=1e-7)
    r_g, r_r = npy(got[1]), ref[1]
    assert got[1].shape == (n, 1)
    assert (r_g == r_r).mean() > 0.995 and np.abs(r_g - r_r).max() <= 1


@pytest.mark.parametrize("n", [3_000, 4_097, 65_537, 100_000, 1_000_003, 4_194_304, 4_194_305])
@pytest.mark.parametrize("dist", ["random", "equal", "two", "sorted", "reversed", "narrow"])
def test_depth_order_sort_is_exact_and_stable(n, dist):
    """gsr_depth_order (purpose-built radix sort between 64k and 4M items, rocPRIM
    outside) == stable argsort by (depth, index); culled splats first; the scan of
    the tile counts follows that order."""
    import rasterizer.cuda as C

    if n > 2_000_000 and dist not in ("random", "equal"):
        pytest.skip("large sizes: two distributions are enough")
    rng = np.random.default_rng(n % 1000 + len(dist))
    if dist == "random":
        d = rng.uniform(0.01, 1000.0, n)
    elif dist == "equal":
        d = np.full(n, 3.25)
    elif dist == "two":
        d = rng.choice([2.0, 7.5], n)
    elif dist == "sorted":
        d = np.linspace(0.5, 50.0, n)
    elif dist == "reversed":
        d = np.linspace(50.0, 0.5, n)
    else:  # all keys share their three high bytes
        d = (np.float32(4.0) + rng.integers(0, 200, n).astype(np.float32) * np.float32(4.7683716e-07))
    d = d.astype(np.float32)
    radii = np.ones(n, np.int32)
    radii[rng.integers(0, n, n // 10)] = 0  # culled: key 0, emitted first
    tiles = rng.integers(0, 5, n).astype(np.int32) * (radii > 0)
    order, cum = C.depth_order(cu(d), cu(radii), cu(tiles))
    key = np.where(radii > 0, d, 0).astype(np.float32)
    ref = np.argsort(key.view(np.uint32), kind="stable")
    assert np.array_equal(npy(order), ref.astype(np.int32))
    assert np.array_equal(npy(cum), np.cumsum(tiles[ref]).astype(np.int32))


def _depth_distributions(n, rng):
    yield "two_octaves", rng.uniform(2.5, 7.5, n)
    yield "seventeen_octaves", rng.uniform(0.01, 1000.0, n)
    yield "equal", np.full(n, 3.25)
    yield "two_values", rng.choice([2.0, 7.5], n)
    yield "sorted", np.linspace(0.5, 50.0, n)
    yield "reversed", np.linspace(50.0, 0.5, n)
    yield "one_bucket", np.float32(4.0) + rng.integers(0, 200, n).astype(np.float32) * np.float32(4.7683716e-07)
    yield "normal", np.abs(rng.normal(5.0, 0.7, n)) + 0.2
    d = rng.uniform(1.0, 100.0, n)
    d[: n // 2] = np.float32(1.0) + rng.integers(0, 400_000, n // 2).astype(np.float32) * np.float32(1.1920929e-07)
    yield "half_in_one_bucket", d
    d = rng.uniform(1.0, 100.0, n)
    d[: n // 2] = 1.5
    yield "half_equal", d
    yield "lognormal", np.exp(rng.normal(1.0, 1.0, n))
    yield "every_octave", np.exp(rng.uniform(-80, 80, n))
    d = rng.uniform(2.0, 6.0, n)
    d[rng.integers(0, n, 40)] = rng.uniform(1e-6, 1e-3, 40)   # a few keys far outside the sampled octaves:
    d[rng.integers(0, n, 40)] = rng.uniform(1e4, 1e9, 40)     # the underflow / overflow buckets
    yield "outliers", d
    yield "by_position", np.sort(rng.uniform(0.3, 30.0, n))[np.argsort(np.arange(n) % 977, kind="stable")]


@pytest.mark.parametrize("n", [4_096, 10_000, 65_537, 300_000, 1_000_003, 1_572_865, 3_000_000, 4_194_304])
@pytest.mark.parametrize("mode", ["bucket", "auto"])
def test_depth_order_without_counts_bucket_sort(n, mode, monkeypatch):
    """The order-only depth sort (lists without counts: the product path at >= 1 M list entries): one bucket pass + one
    in-LDS pass (sort_bucket.hip) == stable argsort by (depth, index), culled splats first, for depth distributions
    that stress the sampled bucket map -- many octaves, one octave, one bucket, clusters of equal keys, outliers
    beyond the sampled octaves, 4096 and 8192 buckets.  `auto` lets the pinned hint send calls back to the four
    LSD passes after a view whose buckets overflowed: the result may never depend on the choice."""
    import rasterizer.cuda as C

    if mode == "auto":
        monkeypatch.delenv("GSR_DEPTH_SORT", raising=False)
    else:
        monkeypatch.setenv("GSR_DEPTH_SORT", "bucket")
    rng = np.random.default_rng(n % 1013)
    for name, d in _depth_distributions(n, rng):
        if n > 2_000_000 and name in ("equal", "two_values", "one_bucket", "half_equal"):
            continue  # (tens of ms each on the slow path; covered at the smaller sizes)
        d = d.astype(np.float32)
        radii = np.ones(n, np.int32)
        radii[rng.integers(0, n, n // 7)] = 0
        order, cum = C.depth_order(cu(d), cu(radii), None)
        assert cum is None
        key = np.where(radii > 0, d, 0).astype(np.float32)
        ref = np.argsort(key.view(np.uint32), kind="stable").astype(np.int32)
        assert np.array_equal(npy(order), ref), (name, n, mode)
        if n <= 1_000_003 and name in ("two_octaves", "normal", "half_equal", "outliers"):
            # with counts (one per Gaussian, and three tile-row bands): gathered where the order is written, then scanned
            for rows in (1, 3):
                tiles = rng.integers(0, 5, (rows, n)).astype(np.int32) * (radii > 0)
                order2, cum2 = C.depth_order(cu(d), cu(radii), cu(tiles.reshape(-1)))
                assert np.array_equal(npy(order2), ref), (name, n, mode, rows)
                assert np.array_equal(npy(cum2), np.cumsum(tiles[:, ref].reshape(-1)).astype(np.int32)), (name, n, rows)


@pytest.mark.parametrize("n,W,H,ck,opac_hi", [(3000, 160, 96, {}, 1.0), (20_000, 317, 203, {"yaw": 0.3}, 1.0),
                                             (5000, 256, 256, {}, 0.02), (200_000, 640, 360, {}, 1.0),
                                             (30_000, 3840, 2160, {}, 1.0), (20_000, 2560, 1440, {"yaw": 0.2}, 1.0)])
def test_exact_lists_drop_only_dead_pairs(n, W, H, ck, opac_hi, monkeypatch):
    """count_reach + bin_sorted(conics, opacities): per tile, the list is a
    subsequence of the reference's; every dropped (Gaussian, tile) pair has
    alpha < 1/255 at every pixel of the tile; the image composited from the short
    lists is bit-identical, gradients equal up to atomic summation order.  The 4K and
    1440p cases run in 4 and 2 tile-row bands (grids above 16384 tiles).  (Bitwise under the single walk: depth
    segments, which cut a tile's list by its LENGTH, are switched off -- their own tests compare to rounding.)"""
    import rasterizer.cuda as C

    monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (1, 0))
    bw = 16
    cam, sc = make(n, W, H, cam_kw=ck, scale_lo=0.01, scale_hi=0.2 if W < 2000 else 0.08)
    rng = np.random.default_rng(n)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    opac = (sc["opacities"] * opac_hi).astype(np.float32)
    if opac_hi < 1.0:
        opac[::7] = 0.0  # never visible
    colors = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    g = dict(xys=cu(xys), depths=cu(depths), radii=cu(radii), conics=cu(conics), tiles=cu(tiles), opac=cu(opac),
             colors=cu(colors))
    # reference lists
    order, cum = C.depth_order(g["depths"], g["radii"], g["tiles"])
    I = int(cum[-1].item())
    ids_ref, bins_ref = C.bin_sorted(n, I, order, cum, g["xys"], g["radii"], tb, bw)
    # exact lists
    # grids above 16384 tiles: two-level partition (one count per Gaussian) AND tile-row bands
    nb = C.tile_bands(tb)
    assert nb == (1 if tb[0] * tb[1] <= 16384 else -(-tb[1] // (8192 // tb[0])))
    if nb > 1:
        cnt1, recs1 = C.count_reach(g["xys"], g["radii"], g["conics"], g["opac"], tb)  # bands = 1
        o1, c1 = C.depth_order(g["depths"], g["radii"], cnt1)
        I1 = int(c1[-1].item())
        two_level = C.bin_sorted(n, I1, o1, c1, g["xys"], g["radii"], tb, bw, recs1)
    cnt_bands, recs = C.count_reach(g["xys"], g["radii"], g["conics"], g["opac"], tb, bands=nb)
    bands = nb
    assert cnt_bands.shape == (bands * n,) and (cnt_bands >= 0).all()
    cnt = cnt_bands.view(bands, n).sum(0).to(torch.int32)
    assert (cnt <= g["tiles"]).all()
    assert (cnt[g["radii"] <= 0] == 0).all()
    order2, cum2 = C.depth_order(g["depths"], g["radii"], cnt_bands)
    assert cum2.shape == (bands * n,)
    assert torch.equal(order, order2)
    I2 = int(cum2[-1].item())
    assert 0 < I2 < I
    ids_ex, bins_ex = C.bin_sorted(n, I2, order2, cum2, g["xys"], g["radii"], tb, bw, recs)
    if nb > 1:  # both large-grid paths build the very same lists
        assert I1 == I2 and torch.equal(two_level[0], ids_ex) and torch.equal(two_level[1], bins_ex)
        cap = torch.zeros(1, dtype=torch.int32).pin_memory()
        ids_c, bins_c = C.bin_sorted(n, I1 + 5000, o1, c1, g["xys"], g["radii"], tb, bw, recs1, device_sized=True,
                                     count_out=cap)
        torch.cuda.synchronize()
        assert int(cap[0]) == I1 and torch.equal(ids_c[:I1], ids_ex) and torch.equal(bins_c, bins_ex)
        ids_s, bins_s = C.bin_sorted(n, I1 // 3, o1, c1, g["xys"], g["radii"], tb, bw, recs1, device_sized=True,
                                     count_out=cap)
        torch.cuda.synchronize()
        assert int(cap[0]) == I1 and int(bins_s.max()) <= I1 // 3  # cut memory-safely
    # device-sized variant: capacity instead of the exact length, count handed back through
    # device-accessible (pinned host) memory; a capacity that is too small cuts the lists
    if True:
        count = torch.zeros(1, dtype=torch.int32).pin_memory()
        ids_cap, bins_cap = C.bin_sorted(n, I2 + 1000, order2, cum2, g["xys"], g["radii"], tb, bw, recs,
                                         device_sized=True, count_out=count)
        torch.cuda.synchronize()
        assert int(count[0]) == I2
        assert torch.equal(ids_cap[:I2], ids_ex) and torch.equal(bins_cap, bins_ex)
        small = I2 // 2
        ids_cut, bins_cut = C.bin_sorted(n, small, order2, cum2, g["xys"], g["radii"], tb, bw, recs,
                                         device_sized=True, count_out=count)
        torch.cuda.synchronize()
        assert int(count[0]) == I2 and int(bins_cut.max()) <= small  # memory-safe, caller rebuilds
    ids_ref_n, bins_ref_n, ids_ex_n, bins_ex_n = npy(ids_ref), npy(bins_ref), npy(ids_ex), npy(bins_ex)
    assert np.array_equal(np.bincount(ids_ex_n, minlength=n), npy(cnt))
    px = np.arange(bw, dtype=np.float32)
    checked = 0
    for t in range(tb[0] * tb[1]):
        a = ids_ref_n[bins_ref_n[t, 0]:bins_ref_n[t, 1]]
        b = ids_ex_n[bins_ex_n[t, 0]:bins_ex_n[t, 1]]
        keep = np.isin(a, b)
        assert np.array_equal(a[keep], b), f"tile {t}: not a subsequence"
        if t % 7 and n > 50_000:
            continue  # the per-pixel check below on a sample of the tiles
        dropped = a[~keep]
        if dropped.size == 0:
            continue
        X = (t % tb[0]) * bw + px[None, None, :]
        Y = (t // tb[0]) * bw + px[None, :, None]
        dx = xys[dropped, 0][:, None, None] - X
        dy = xys[dropped, 1][:, None, None] - Y
        cn = conics[dropped].astype(np.float64)
        sigma = 0.5 * (cn[:, 0, None, None] * dx * dx + cn[:, 2, None, None] * dy * dy) + cn[:, 1, None, None] * dx * dy
        alpha = np.minimum(0.999, opac[dropped, 0][:, None, None] * np.exp(-sigma))
        assert (alpha[sigma >= 0] < 1.0 / 255.0).all(), f"tile {t}: a live pair was dropped"
        checked += dropped.size
    assert checked > 0
    # compositing from either list
    block, img_size = (bw, bw, 1), (W, H, 1)
    bg = cu(np.array(S.BACKGROUND, np.float32))
    img_r, T_r, idx_r = C.rasterize_forward(tb, block, img_size, ids_ref, bins_ref, g["xys"], g["conics"],
                                            g["colors"], g["opac"], bg)
    img_e, T_e, idx_e = C.rasterize_forward(tb, block, img_size, ids_ex, bins_ex, g["xys"], g["conics"],
                                            g["colors"], g["opac"], bg)
    assert torch.equal(img_r, img_e) and torch.equal(T_r, T_e)
    v_img = cu(rng.standard_normal((H, W, 3)).astype(np.float32))
    v_alpha = cu(rng.standard_normal((H, W)).astype(np.float32))
    gr = C.rasterize_backward(H, W, bw, ids_ref, bins_ref, g["xys"], g["conics"], g["colors"], g["opac"], bg,
                              T_r, idx_r, v_img, v_alpha)
    ge = C.rasterize_backward(H, W, bw, ids_ex, bins_ex, g["xys"], g["conics"], g["colors"], g["opac"], bg,
                              T_e, idx_e, v_img, v_alpha)
    for a, b, nm in zip(gr, ge, ("v_xy", "v_conic", "v_colors", "v_opacity")):
        a, b = npy(a), npy(b)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max() + 1e-12, nm


@pytest.mark.parametrize("n,W,H,scale_hi", [(20_000, 272, 16368, 0.08),   # 17 x 1023 tiles: the row tables at their largest
                                            (20_000, 16384, 272, 0.08),   # 1024 x 17: the 1024-column scatter
                                            (6_000, 3200, 1800, 1.5),     # huge splats: batches above the start-mask size
                                            (70_000, 3200, 1800, 0.02)])  # several scan groups, mostly one-tile splats
def test_two_level_partition_equals_banded_lists(n, W, H, scale_hi):
    """The two large-grid list builders (csrc/tile_partition2.hip and the tile-row bands of
    csrc/tile_scatter.hip) produce the same ids / bins, bit for bit, on grid shapes and splat
    sizes that reach the corners of the two-level code."""
    import rasterizer.cuda as C

    bw = 16
    cam, sc = make(n, W, H, scale_lo=0.01, scale_hi=scale_hi)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    nb = C.tile_bands(tb)
    assert nb > 1
    g = dict(xys=cu(xys), depths=cu(depths), radii=cu(radii), conics=cu(conics), opac=cu(sc["opacities"]))
    cnt1, recs1 = C.count_reach(g["xys"], g["radii"], g["conics"], g["opac"], tb)
    o1, c1 = C.depth_order(g["depths"], g["radii"], cnt1)
    I1 = int(c1[-1].item())
    assert I1 > 0
    ids_t, bins_t = C.bin_sorted(n, I1, o1, c1, g["xys"], g["radii"], tb, bw, recs1)
    cntb, recs = C.count_reach(g["xys"], g["radii"], g["conics"], g["opac"], tb, bands=nb)
    o2, c2 = C.depth_order(g["depths"], g["radii"], cntb)
    I2 = int(c2[-1].item())
    ids_b, bins_b = C.bin_sorted(n, I2, o2, c2, g["xys"], g["radii"], tb, bw, recs)
    assert I1 == I2 and torch.equal(bins_t, bins_b) and torch.equal(ids_t, ids_b)
    assert int(bins_t[:, 1].max()) == I1


@pytest.mark.parametrize("n,W,H,lo,hi", [(30_000, 3840, 2160, 0.01, 0.08),     # large grid: always the two-level path
                                         (300_000, 1920, 1080, 0.01, 0.08),    # 1080p with long lists
                                         (1_000_000, 1920, 1080, 0.0025, 0.025)])  # the bench default's sizes
def test_lists_without_counts_equal_the_full_flow(n, W, H, lo, hi):
    """include/gsraster.h "Lists without counts": records only + order only + cum_sorted = NULL
    build the same ids / bins and report the same number of entries as the full flow."""
    import rasterizer.cuda as C

    bw = 16
    cam, sc = make(n, W, H, scale_lo=lo, scale_hi=hi)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    g = dict(xys=cu(xys), depths=cu(depths), radii=cu(radii), conics=cu(conics), opac=cu(sc["opacities"]))
    cnt, recs = C.count_reach(g["xys"], g["radii"], g["conics"], g["opac"], tb)
    order, cum = C.depth_order(g["depths"], g["radii"], cnt)
    I = int(cum[-1].item())
    cap = int(1.3 * I)
    assert not C.lists_need_counts(n, cap, tb, device_sized=True)
    assert C.lists_need_counts(n, cap, tb, device_sized=True, want_slots=True)  # slots: the single-pass / banded path
    ids_full, bins_full = C.bin_sorted(n, I, order, cum, g["xys"], g["radii"], tb, bw, recs)
    none, recs2 = C.count_reach(g["xys"], g["radii"], g["conics"], g["opac"], tb, counts=False)
    assert none is None and torch.equal(recs, recs2)
    order2, cum2 = C.depth_order(g["depths"], g["radii"], None)
    assert cum2 is None and torch.equal(order, order2)
    # ... and as ONE call (the records ride in the bucket sort's first launch above 64 k Gaussians)
    recs3, order3 = C.reach_records_depth_order(g["xys"], g["radii"], g["conics"], g["opac"], g["depths"], tb)
    assert torch.equal(recs3, recs) and torch.equal(order3, order)
    recs4, order4 = C.reach_records_depth_order(g["xys"], g["radii"], g["conics"], g["opac"], g["depths"], tb,
                                                extra_rows=1)
    assert torch.equal(recs4[:n], recs) and not recs4[n:].any() and torch.equal(order4, order)
    count = torch.zeros(1, dtype=torch.int32).pin_memory()
    ids_lean, bins_lean = C.bin_sorted(n, cap, order2, None, g["xys"], g["radii"], tb, bw, recs2, device_sized=True,
                                       count_out=count)
    torch.cuda.synchronize()
    assert int(count[0]) == I
    assert torch.equal(bins_lean, bins_full) and torch.equal(ids_lean[:I], ids_full)
    # a list too short for the two-level path must not be asked for without counts
    if W < 3000:
        assert C.lists_need_counts(n, 100_000, tb, device_sized=True)
        with pytest.raises(RuntimeError, match="cum_sorted may be NULL only"):
            C.bin_sorted(n, 100_000, order2, None, g["xys"], g["radii"], tb, bw, recs2, device_sized=True,
                         count_out=count)


def test_list_builders_agree_on_random_shapes():
    """tools/exp/fuzz_lists.py in a process of its own (GSR_TILE_SORT is read once per process): the
    two-level partition forced onto every size -- 1 to 400 k Gaussians, 1 x 1 to 1024 x 17 tile grids,
    sub-pixel to screen-filling splats -- builds the lists of the single-pass / banded scatter, with
    and without counts, and cuts memory-safely at a capacity that is too small."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSR_TILE_SORT="t")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_lists.py"), "60", "11"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") >= 40


def test_compositing_kernels_on_random_shapes():
    """tools/exp/fuzz_raster.py: tile16 (through the _ex entry points) and generic compositing kernels against
    the oracle on 60 random image shapes (1 x 1 px up), block widths, splat sizes and opacities; gradients
    within 1e-3 of the largest + 2e-5 of the summed term magnitudes on Gaussians without a borderline pixel."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_raster.py"), "60", "31"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") >= 40
    # once more with the depth-segment kernels forced onto short lists (the fuzz's grids are all below the 1 100
    # tiles on which every tile above 96 entries is split): 5 runs for every list of more than one chunk
    env = dict(os.environ, GSR_TUNE='{"depth_segments": 5, "depth_segments_min": 64, "depth_segments_fwd": 0}')
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_raster.py"), "40", "57"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") >= 25


def test_projection_and_sh_on_random_cameras_and_scales():
    """tools/exp/fuzz_project.py: 60 random cameras (roll, translation), image shapes from 1 x 1, block
    widths, scales from 1e-4 to 30, points behind and at the near plane, global scale and clip threshold:
    the forward projection stays bit-identical to the oracle, the backward within 1e-3 per row, SH 1e-5."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_project.py"), "60", "41"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") == 60


def test_depth_order_on_random_sizes_rows_and_key_distributions():
    """tools/exp/fuzz_depth_order.py: 1 to 4 M + 1 Gaussians (both sort paths), 1-16 count rows, ties / narrow /
    sorted / reversed keys, culled Gaussians: the stable order and the inclusive prefix of the counts in that
    order (the decoupled look-back scan) against numpy; also the order-only call."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_depth_order.py"), "40", "81"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") == 40


def test_count_reach_errors():
    import rasterizer.cuda as C

    z = torch.zeros
    with pytest.raises(RuntimeError):
        C.count_reach(z(4, 2, device=DEV), z(4, dtype=torch.int32, device=DEV), z(4, 3, device=DEV),
                      z(3, 1, device=DEV), (4, 4, 1))
    with pytest.raises(RuntimeError):  # the reach test is defined on 16x16 tiles only
        C.bin_sorted(4, 4, z(4, dtype=torch.int32, device=DEV), z(4, dtype=torch.int32, device=DEV),
                     z(4, 2, device=DEV), z(4, dtype=torch.int32, device=DEV), (4, 4, 1), 8,
                     z(4, 32, dtype=torch.uint8, device=DEV))
    with pytest.raises(RuntimeError):  # wrong record buffer size
        C.bin_sorted(4, 4, z(4, dtype=torch.int32, device=DEV), z(4, dtype=torch.int32, device=DEV),
                     z(4, 2, device=DEV), z(4, dtype=torch.int32, device=DEV), (4, 4, 1), 16,
                     z(3, 32, dtype=torch.uint8, device=DEV))
    out, _ = C.count_reach(z(0, 2, device=DEV), z(0, dtype=torch.int32, device=DEV), z(0, 3, device=DEV),
                        z(0, 1, device=DEV), (4, 4, 1))
    assert out.numel() == 0


@pytest.mark.parametrize("rgbd", [False, True])
def test_deep_tiles_split_over_four_waves_give_identical_results(rgbd, monkeypatch):
    """A tile whose list is longer than the deep-tile threshold is composited by four waves,
    one per 8x8 sub-tile (raster_common.h): forward outputs are bit-identical to the
    one-wave-per-tile walk, gradients equal up to the order of the float atomics."""
    import rasterizer.cuda as C

    n, W, H, bw = 60_000, 320, 208, 16
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=5, scale_lo=0.01, scale_hi=0.08, longtail=True)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    rng = np.random.default_rng(2)
    colors = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    order, cum = C.depth_order(cu(depths), cu(radii), cu(tiles))
    I = int(cum[-1].item())
    ids, bins = C.bin_sorted(n, I, order, cum, cu(xys), cu(radii), tb, bw)
    lens = (bins[:, 1] - bins[:, 0]).cpu().numpy()
    assert lens.max() > 4 * np.median(lens)  # a long tail
    bg = cu(np.array(S.BACKGROUND, np.float32))
    v_img = torch.rand(H, W, 3, device=DEV) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV) * 2 - 1
    v_ext = torch.rand(H, W, device=DEV) * 2 - 1

    monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (1, 0))  # (their own tests follow)

    def run(threshold):
        monkeypatch.setattr(C, "deep_tile_threshold", lambda entries, num_tiles, backward=False: threshold)
        if rgbd:
            f = C.rasterize_forward_rgbd(tb, (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors), cu(depths),
                                         cu(sc["opacities"]), bg, 0.0)
            b = C.rasterize_backward_rgbd(H, W, ids, bins, cu(xys), cu(conics), cu(colors), cu(depths),
                                          cu(sc["opacities"]), bg, 0.0, f[2], f[3], v_img, v_ext, v_alpha)
        else:
            f = C.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors),
                                    cu(sc["opacities"]), bg)
            b = C.rasterize_backward(H, W, bw, ids, bins, cu(xys), cu(conics), cu(colors), cu(sc["opacities"]), bg,
                                     f[1], f[2], v_img, v_alpha)
        return f, b

    f0, b0 = run(0)                                   # one wave per tile
    thr = int(np.percentile(lens, 70))               # 30 % of the tiles split
    f1, b1 = run(thr)
    f2, b2 = run(1)                                   # every non-trivial tile split
    for fa in (f1, f2):
        for x, y in zip(f0, fa):
            assert torch.equal(x, y)
    for ba in (b1, b2):
        for x, y in zip(b0, ba):
            scale = x.abs().max().item()
            assert (x - y).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("opaque,rgbd", [(False, False), (True, False), (True, True)])
def test_depth_segments_of_the_forward_equal_the_single_walk(opaque, rgbd, monkeypatch):
    """gsr_rasterize_forward_seg: the lists of the split tiles cut into 2 .. 16 runs; a pre-pass gives every run the
    transmittance in front of it, the runs composite in parallel with the unchanged stop rule, a combine pass adds
    them up in list order.  Image and final T equal the single walk's to rounding; the last drawn index is the same
    wherever the pixel's decisions are not within rounding of a threshold (all but a handful of pixels).  Long-tail
    scene, and the same scene nearly opaque (every pixel finishes inside some run)."""
    import rasterizer.cuda as C

    n, W, H, bw = 60_000, 320, 208, 16
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=5, scale_lo=0.01, scale_hi=0.08, longtail=True)
    opac = sc["opacities"].copy()
    if opaque:
        opac = np.maximum(opac, 0.97).astype(np.float32)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    colors = np.random.default_rng(2).uniform(0, 1, (n, 3)).astype(np.float32)
    order, cum = C.depth_order(cu(depths), cu(radii), cu(tiles))
    I = int(cum[-1].item())
    ids, bins = C.bin_sorted(n, I, order, cum, cu(xys), cu(radii), tb, bw)
    lens = (bins[:, 1] - bins[:, 0]).cpu().numpy()
    bg = cu(np.array(S.BACKGROUND, np.float32))

    monkeypatch.setattr(C, "_segment_knobs", lambda: (16, 1100, 512, 0))  # (no separate cap on the forward's runs)

    def run(segs, least, threshold=96, ex=True):
        monkeypatch.setattr(C, "deep_tile_threshold", lambda entries, num_tiles, backward=False: threshold)
        monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (segs, least))
        if rgbd:  # the fourth channel (the depths) rides in the colour image's fourth plane for the comparison
            img, ext, T, idx, a = C.rasterize_forward_rgbd(tb, (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors),
                                                           cu(depths), cu(opac), bg, 0.25, want_alpha=True)
            out = (torch.cat([img, ext[..., None] / float(depths.max())], -1), T, idx, a)
            return out if ex else out[:3]
        if ex:
            return C.rasterize_forward_ex(tb, (bw, bw, 1), (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors),
                                          cu(opac), bg, want_alpha=True)
        return C.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors), cu(opac), bg)

    img0, T0, idx0, a0 = run(1, 0)
    if opaque:
        assert (T0 < 1e-3).float().mean().item() > 0.5
    for segs, least, thr in ((2, 64, 96), (5, 64, 96), (8, 512, 96), (16, 64, 1), (8, 64, int(np.percentile(lens, 70)))):
        img, T, idx, a = run(segs, least, thr)
        assert (img - img0).abs().max().item() <= 2e-6, (segs, least, thr)
        assert (T - T0).abs().max().item() <= 1e-6 and torch.equal(a, 1 - T)
        assert (idx != idx0).float().mean().item() <= 1e-3, (segs, least, thr)
        img_b, T_b, idx_b = run(segs, least, thr, ex=False)  # (the plain entry takes the same route)
        assert torch.equal(img_b, img) and torch.equal(T_b, T) and torch.equal(idx_b, idx)
    with pytest.raises(RuntimeError):
        run(17, 64)


@pytest.mark.parametrize("opaque,rgbd", [(False, False), (True, False), (True, True)])
def test_depth_segments_of_the_backward_equal_the_single_walk(opaque, rgbd, monkeypatch):
    """gsr_rasterize_backward_seg: the lists of the split tiles cut into 2 .. 16 runs, every run walked by its own
    waves from the state a pre-pass computed (T and the colour buffer are mapped affinely by a run).  Gradients equal
    the single walk's to rounding: a long-tail scene (lists of several thousand entries next to short ones; runs
    behind every pixel's last index; tiles below the minimum stay whole), and the same scene nearly opaque -- pixels
    saturate after a few splats, and alpha exceeds the backward's 0.99 clamp, where the backward's T is not the
    forward's (backward.cu:133-303 restated in raster_bwd.hip)."""
    import rasterizer.cuda as C

    n, W, H, bw = 60_000, 320, 208, 16
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=5, scale_lo=0.01, scale_hi=0.08, longtail=True)
    opac = sc["opacities"].copy()
    if opaque:
        opac = np.maximum(opac, 0.97).astype(np.float32)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    colors = np.random.default_rng(2).uniform(0, 1, (n, 3)).astype(np.float32)
    order, cum = C.depth_order(cu(depths), cu(radii), cu(tiles))
    I = int(cum[-1].item())
    ids, bins = C.bin_sorted(n, I, order, cum, cu(xys), cu(radii), tb, bw)
    lens = (bins[:, 1] - bins[:, 0]).cpu().numpy()
    assert lens.max() > 2000 and np.median(lens) < 1000
    bg = cu(np.array(S.BACKGROUND, np.float32))
    v_img = torch.rand(H, W, 3, device=DEV) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV) * 2 - 1
    v_ext = torch.rand(H, W, device=DEV) * 2 - 1
    monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (1, 0))
    f = C.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors), cu(opac), bg)
    if opaque:
        assert (f[1] < 1e-3).float().mean().item() > 0.5  # most pixels saturated

    def run(segs, least, threshold=96):
        monkeypatch.setattr(C, "deep_tile_threshold", lambda entries, num_tiles, backward=False: threshold)
        monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (segs, least))
        if rgbd:
            return C.rasterize_backward_rgbd(H, W, ids, bins, cu(xys), cu(conics), cu(colors), cu(depths), cu(opac), bg,
                                             0.25, f[1], f[2], v_img, v_ext, v_alpha)
        return C.rasterize_backward(H, W, bw, ids, bins, cu(xys), cu(conics), cu(colors), cu(opac), bg, f[1], f[2],
                                    v_img, v_alpha)

    ref = run(1, 0)
    assert all(torch.isfinite(t).all() for t in ref) and ref[0].abs().max().item() > 0
    for segs, least, thr in ((2, 64, 96), (5, 64, 96), (8, 512, 96), (16, 64, 1), (8, 64, int(np.percentile(lens, 70)))):
        got = run(segs, least, thr)
        for x, y in zip(ref, got):
            assert (x - y).abs().max().item() <= 2e-5 * x.abs().max().item(), (segs, least, thr)
    with pytest.raises(RuntimeError):
        run(17, 64)  # at most 16 runs


@pytest.mark.parametrize("tail,threshold", [(0, 300), (8, 300), (16, 10**6), (63, 40)])
def test_job_order_is_a_partition_longest_first_and_changes_no_result(tail, threshold, monkeypatch):
    """GSR_DEEP_ORDERED (round 5): the compositing entries build, behind tile_bins, the order in which their workgroups
    take jobs -- per XCD, every tile exactly once (whole, or as its four sub-tile jobs), by decreasing half-octave bucket
    of the list length (a split tile's jobs keyed by it times the measured cost ratio of a sub-tile wave), stable inside a bucket, the last `tail`/64 of the
    whole-tile jobs cut into sub-tile jobs behind everything else.  Forward images and backward gradients are those of
    the static block order (bit for bit / to the float atomics' summation order)."""
    import rasterizer.cuda as C

    n, W, H, bw = 60_000, 640, 368, 16
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=9, scale_lo=0.005, scale_hi=0.05, longtail=True)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    nt = tb[0] * tb[1]
    colors = np.random.default_rng(2).uniform(0, 1, (n, 3)).astype(np.float32)
    order, cum = C.depth_order(cu(depths), cu(radii), cu(tiles))
    ids, bins = C.bin_sorted(n, int(cum[-1].item()), order, cum, cu(xys), cu(radii), tb, bw)
    lens = (bins[:, 1] - bins[:, 0]).cpu().numpy()
    bg = cu(np.array(S.BACKGROUND, np.float32))
    a = (tb, (bw, bw, 1), (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors), cu(sc["opacities"]), bg)
    monkeypatch.setattr(C, "deep_tile_threshold", lambda entries, num_tiles, backward=False: threshold)
    monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (1, 0))
    C._order_knob()
    C._deep_knobs()
    monkeypatch.setitem(C._deep_cache, "v", (1.2, 256, 0, 96, 0, 2.0, False))  # (no "small grid": this one has 920 tiles)
    monkeypatch.setitem(C._order_cache, "v", False)
    f0 = C.rasterize_forward_ex(*a, want_alpha=True)
    v_img = torch.rand(H, W, 3, device=DEV) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV) * 2 - 1
    b0 = C.rasterize_backward(H, W, bw, *a[3:], f0[1], f0[2], v_img, v_alpha)
    monkeypatch.setitem(C._order_cache, "v", True)
    monkeypatch.setitem(C._order_cache, "tail", tail)
    monkeypatch.setitem(C._order_cache, "tail_bwd", tail)
    assert C.deep_arg(bins, ids.numel(), nt, tile_bounds=tb) & C.GSR_DEEP_ORDERED
    f1 = C.rasterize_forward_ex(*a, want_alpha=True)
    for x, y in zip(f0, f1):
        assert torch.equal(x, y)
    first = (C.tile_jobs_ints(tb) - 2) // 2  # the forward's array: the first of the two behind tile_bins
    jobs = torch.empty(0, dtype=torch.int32, device=DEV).set_(bins.untyped_storage(), bins.storage_offset() + 2 * nt,
                                                               (first,)).cpu().numpy().reshape(-1, 8)
    b1 = C.rasterize_backward(H, W, bw, *a[3:], f0[1], f0[2], v_img, v_alpha)
    for x, y in zip(b0, b1):
        assert (x - y).abs().max().item() <= 2e-5 * x.abs().max().item()
    seen = np.zeros(nt, np.int32)
    for xcd in range(8):
        col = jobs[:, xcd]
        live = col[col >= 0]
        tile, allowed = live & ((1 << 27) - 1), live >> 27
        np.add.at(seen, tile, allowed)
        whole = allowed == 15
        deep = lens[tile] > threshold
        assert np.all(deep[~whole] | (lens[tile][~whole] > 0))  # sub-tile jobs: deep tiles, or non-empty tail tiles
        # the sorted part (before the tail's sub-tile jobs of shallow tiles): non-increasing half-octave buckets
        # (a split tile's jobs are keyed by its length x the measured cost ratio of a sub-tile wave, pinned to 1/8
        #  here through GSR_DEEP_SPLIT_KEY's C-side default when nothing has been measured ... the ratio moves with
        #  the launches of this very process, so only the two classes' own orders are asserted)
        q = np.where(lens[tile] > 0, np.floor(2 * np.log2(np.maximum(lens[tile], 1)) + 1e-9), -1)
        body = whole | deep
        assert np.all(np.diff(q[whole]) <= 0), (xcd, q[whole][:40])
        # (split jobs: by half-octaves of length x ratio -- whatever the ratio, a later job's tile is never more than
        #  one half-octave longer than an earlier one's)
        ls = lens[tile][deep & ~whole].astype(np.float64)
        assert np.all(ls[1:] <= ls[:-1] * 1.4143), (xcd, ls[:40])
        n_tail_tiles = int((~body).sum()) // 4
        n_whole_total = int(whole.sum()) + n_tail_tiles
        assert n_tail_tiles <= (n_whole_total * tail) // 64
        if tail == 0:
            assert n_tail_tiles == 0
    assert np.all(seen == 15), "every tile exactly once: whole (15) or its four sub-tiles (1 + 2 + 4 + 8)"


def test_lists_that_are_all_alike_keep_the_backward_unsplit(monkeypatch):
    """The order kernel's guard (raster_common.h, `alike`): when the longest list is within 1.5 x the mean of the
    non-empty ones -- a random cloud -- the backward's order holds whole-tile jobs only, whatever the threshold says
    (the grid-scaled threshold of small grids is below such a scene's mean), while the forward's order splits as told;
    results are those of the static order."""
    import rasterizer.cuda as C

    n, W, H, bw = 150_000, 640, 368, 16
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=11, scale_lo=0.004, scale_hi=0.02)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    nt = tb[0] * tb[1]
    colors = np.random.default_rng(2).uniform(0, 1, (n, 3)).astype(np.float32)
    order, cum = C.depth_order(cu(depths), cu(radii), cu(tiles))
    ids, bins = C.bin_sorted(n, int(cum[-1].item()), order, cum, cu(xys), cu(radii), tb, bw)
    lens = (bins[:, 1] - bins[:, 0]).cpu().numpy()
    assert lens.max() <= 1.5 * lens[lens > 0].mean(), (lens.max(), lens.mean())  # the premise: lists all alike
    threshold = int(0.6 * lens.mean())
    assert (lens > threshold).mean() > 0.9
    bg = cu(np.array(S.BACKGROUND, np.float32))
    a = (tb, (bw, bw, 1), (W, H, 1), ids, bins, cu(xys), cu(conics), cu(colors), cu(sc["opacities"]), bg)
    monkeypatch.setattr(C, "deep_tile_threshold", lambda entries, num_tiles, backward=False: threshold)
    monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (1, 0))
    C._order_knob()
    C._deep_knobs()
    monkeypatch.setitem(C._deep_cache, "v", (1.2, 256, 0, 96, 0, 2.0, False))
    monkeypatch.setitem(C._order_cache, "v", False)
    f0 = C.rasterize_forward_ex(*a, want_alpha=True)
    v_img = torch.rand(H, W, 3, device=DEV) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV) * 2 - 1
    b0 = C.rasterize_backward(H, W, bw, *a[3:], f0[1], f0[2], v_img, v_alpha)
    monkeypatch.setitem(C._order_cache, "v", True)
    monkeypatch.setitem(C._order_cache, "tail", 0)
    monkeypatch.setitem(C._order_cache, "tail_bwd", 0)
    monkeypatch.setitem(C._order_cache, "grid", 0)
    f1 = C.rasterize_forward_ex(*a, want_alpha=True)
    for x, y in zip(f0, f1):
        assert torch.equal(x, y)
    b1 = C.rasterize_backward(H, W, bw, *a[3:], f0[1], f0[2], v_img, v_alpha)
    for x, y in zip(b0, b1):
        assert (x - y).abs().max().item() <= 2e-5 * x.abs().max().item()
    half = (C.tile_jobs_ints(tb) - 2) // 2
    both = torch.empty(0, dtype=torch.int32, device=DEV).set_(bins.untyped_storage(), bins.storage_offset() + 2 * nt,
                                                               (2 * half,)).cpu().numpy()
    for which, arr in (("forward", both[:half]), ("backward", both[half:])):
        live = arr[arr >= 0]
        tile, allowed = live & ((1 << 27) - 1), live >> 27
        seen = np.zeros(nt, np.int32)
        np.add.at(seen, tile, allowed)
        assert np.all(seen == 15), which
        if which == "backward":
            assert np.all(allowed == 15), "the backward of lists that are all alike runs whole tiles only"
        else:
            assert (allowed != 15).mean() > 0.9  # the forward splits as the threshold says


@pytest.mark.parametrize("segs,least", [(1, 0), (5, 64), (16, 64)])
@pytest.mark.parametrize("opaque", [False, True])
@pytest.mark.parametrize("rgbd", [False, True])
def test_depth_segment_kernels_against_the_oracle(segs, least, opaque, rgbd, monkeypatch):
    """The depth-segment kernels (gsr_rasterize_forward_seg / _backward_seg, plain and RGBD) against the ORACLE, not
    against this package's single walk: the deep-tile threshold forced low (every tile with more than 96 entries is
    split over four waves, the precondition of the runs), 1 / 5 / 16 runs, a translucent long-tail scene and the same
    scene nearly opaque.  Forward: 1e-4 abs on decision-stable pixels, the last drawn index equal there (a run
    boundary may move it at a handful of pixels whose stop decision is within rounding: <= 1e-3 of them);
    backward: 1e-3 relative (floor 1e-3 of the largest), from the ORACLE's forward state."""
    import rasterizer.cuda as C

    n, W, H, bw = 30_000, 320, 208, 16
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=15, scale_lo=0.01, scale_hi=0.08, longtail=True)
    opac = sc["opacities"].copy()
    if opaque:
        opac = np.maximum(opac, 0.97).astype(np.float32)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    I, cum = O.compute_cumulative_intersects(tiles)
    _, _, ks, vs, bins = O.bin_and_sort_gaussians(n, I, xys, depths, radii, cum, tb, bw)
    assert (bins[:, 1] - bins[:, 0]).max() > 1500  # lists long enough for 16 runs of whole chunks
    rng = np.random.default_rng(3)
    colors = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    bg = rng.uniform(0, 1, 3).astype(np.float32)
    ebg = 0.25
    ref = O.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), vs, bins, xys, conics, colors, opac, bg, ambig_eps=1e-5)
    dcol = np.repeat(depths[:, None], 3, 1).astype(np.float32)
    refd = O.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), vs, bins, xys, conics, dcol, opac,
                               np.full(3, ebg, np.float32)) if rgbd else None

    monkeypatch.setattr(C, "_segment_knobs", lambda: (16, 1100, 512, 0))
    monkeypatch.setattr(C, "deep_tile_threshold", lambda entries, num_tiles, backward=False: 96)
    monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (segs, least))
    g = dict(ids=cu(vs), bins=cu(bins), xys=cu(xys), conics=cu(conics), colors=cu(colors), opac=cu(opac), bg=cu(bg))
    if rgbd:
        img, ext, Ts, idx, alpha = C.rasterize_forward_rgbd(tb, (W, H, 1), g["ids"], g["bins"], g["xys"], g["conics"],
                                                            g["colors"], cu(depths), g["opac"], g["bg"], ebg,
                                                            want_alpha=True)
    else:
        img, Ts, idx, alpha = C.rasterize_forward_ex(tb, (bw, bw, 1), (W, H, 1), g["ids"], g["bins"], g["xys"],
                                                     g["conics"], g["colors"], g["opac"], g["bg"], want_alpha=True)
    ok = ~ref[3]
    check_image(npy(img), npy(Ts), ref, ref[3])
    assert (npy(idx)[ok] != ref[2][ok]).mean() <= 1e-3
    assert np.abs(npy(alpha) - (1 - ref[1]))[ok].max() < 1e-4
    if rgbd:
        assert np.abs(npy(ext) - refd[0][..., 0])[ok].max() < 1e-4 * max(1.0, float(depths.max()))
    if opaque:
        assert (ref[1] < 1e-3).mean() > 0.5  # most pixels saturate: the runs behind them are never composited

    # backward from the oracle's forward state
    v_img = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
    v_alpha = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    v_ext = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    rb = list(O.rasterize_backward(H, W, bw, vs, bins, xys, conics, colors, opac, bg, ref[1], ref[2], v_img, v_alpha))
    if rgbd:
        ve3 = np.zeros((H, W, 3), np.float32)
        ve3[..., 0] = v_ext
        rd = O.rasterize_backward(H, W, bw, vs, bins, xys, conics, dcol, opac, np.full(3, ebg, np.float32), refd[1],
                                  refd[2], ve3, np.zeros((H, W), np.float32))
        rb[0], rb[1], rb[3] = rb[0] + rd[0], rb[1] + rd[1], rb[3] + rd[3]
        rb.append(rd[2][:, 0])
        got = C.rasterize_backward_rgbd(H, W, g["ids"], g["bins"], g["xys"], g["conics"], g["colors"], cu(depths),
                                        g["opac"], g["bg"], ebg, cu(ref[1]), cu(ref[2]), cu(v_img), cu(v_ext),
                                        cu(v_alpha))
        names = ["v_xy", "v_conic", "v_colors", "v_extra", "v_opacity"]
        rb = [rb[0], rb[1], rb[2], rb[4], rb[3]]
    else:
        got = C.rasterize_backward(H, W, bw, g["ids"], g["bins"], g["xys"], g["conics"], g["colors"], g["opac"],
                                   g["bg"], cu(ref[1]), cu(ref[2]), cu(v_img), cu(v_alpha))
        names = ["v_xy", "v_conic", "v_colors", "v_opacity"]
    for a, r, nm in zip(got, rb, names):
        grad_close(npy(a).reshape(np.asarray(r).shape), np.asarray(r), name=f"{nm} segs={segs}")


@pytest.mark.parametrize("W,H,n", [(320, 208, 40_000), (3840, 2160, 30_000)])
def test_deterministic_backward_is_bit_reproducible_and_equals_the_atomic_one(W, H, n):
    """gsr_rasterize_backward_det: per-(tile, entry) partials summed per Gaussian in a fixed
    order.  Two runs give bit-identical gradients (the atomic version differs by a few ulp from
    run to run), equal to the atomic result up to that summation order; RGB and RGB + depth; one
    band and four bands (4K)."""
    import rasterizer.cuda as C

    bw = 16
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=8, scale_lo=0.01, scale_hi=0.08 if W < 2000 else 0.05)
    cov3d, xys, depths, radii, conics, comp, tiles = project_cpu(cam, sc, bw)
    tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    g = dict(xys=cu(xys), depths=cu(depths), radii=cu(radii), conics=cu(conics), opac=cu(sc["opacities"]))
    cnt, recs = C.count_reach(g["xys"], g["radii"], g["conics"], g["opac"], tb, bands=C.tile_bands(tb))
    order, cum = C.depth_order(g["depths"], g["radii"], cnt)
    I = int(cum[-1].item())
    ids, bins, slots = C.bin_sorted(n, I + 4096, order, cum, g["xys"], g["radii"], tb, bw, recs, device_sized=True,
                                    want_slots=True)
    # the inverse map: entry e went to slot slots[e]; it is a permutation of [0, I)
    assert torch.equal(torch.sort(slots[:I]).values, torch.arange(I, device=DEV, dtype=torch.int32))
    # ... and entry e of Gaussian order[i] really holds that Gaussian
    owner = torch.repeat_interleave(order.repeat(C.tile_bands(tb)).long(),
                                    torch.diff(cum.long(), prepend=torch.zeros(1, device=DEV, dtype=torch.long)))
    assert torch.equal(ids[slots[:I].long()].long(), owner)
    rng = np.random.default_rng(1)
    colors = cu(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    bg = cu(np.array(S.BACKGROUND, np.float32))
    v_img = torch.rand(H, W, 3, device=DEV) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV) * 2 - 1
    v_ext = torch.rand(H, W, device=DEV) * 2 - 1
    img, Ts, idx = C.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), ids, bins, g["xys"], g["conics"], colors, g["opac"], bg)
    args = (H, W, ids, bins, g["xys"], g["conics"], colors, g["opac"], bg, Ts, idx, v_img, v_alpha, order, cum, slots)
    d1 = C.rasterize_backward_det(*args)
    d2 = C.rasterize_backward_det(*args)
    a = C.rasterize_backward(H, W, bw, ids, bins, g["xys"], g["conics"], colors, g["opac"], bg, Ts, idx, v_img, v_alpha)
    for x, y, z in zip(d1, d2, a):
        assert torch.equal(x, y)                                                    # reproducible bit for bit
        assert (x - z).abs().max().item() <= 2e-5 * z.abs().max().item() + 1e-12    # same sums
    # RGB + extra channel
    f = C.rasterize_forward_rgbd(tb, (W, H, 1), ids, bins, g["xys"], g["conics"], colors, g["depths"], g["opac"], bg, 0.0)
    e1 = C.rasterize_backward_det(H, W, ids, bins, g["xys"], g["conics"], colors, g["opac"], bg, f[2], f[3], v_img,
                                  v_alpha, order, cum, slots, extra=g["depths"], extra_background=0.0,
                                  v_output_extra=v_ext)
    e2 = C.rasterize_backward_det(H, W, ids, bins, g["xys"], g["conics"], colors, g["opac"], bg, f[2], f[3], v_img,
                                  v_alpha, order, cum, slots, extra=g["depths"], extra_background=0.0,
                                  v_output_extra=v_ext)
    b = C.rasterize_backward_rgbd(H, W, ids, bins, g["xys"], g["conics"], colors, g["depths"], g["opac"], bg, 0.0,
                                  f[2], f[3], v_img, v_ext, v_alpha)
    for x, y, z in zip(e1, e2, b):
        assert torch.equal(x, y)
        assert (x - z.view_as(x)).abs().max().item() <= 2e-5 * z.abs().max().item() + 1e-12


@pytest.mark.parametrize("rgbd", [False, True])
def test_nan_cotangents_at_undrawn_pixels_are_never_read(rgbd):
    """The models' depth image is `where(alpha > 0, depth / alpha, max)` (vanilla_gs.py:855): its backward
    hands 0/0 = NaN cotangents to exactly the pixels where nothing was drawn.  The reference's kernel
    branches on `valid` and never reads them (backward.cu:133-303); neither may the select-based tile16
    kernels: gradients with NaN there equal the gradients with 0 there, bit for bit (deterministic mode)."""
    import rasterizer.cuda as C

    n, W, H, bw = 1500, 192, 128, 16
    # a sparse scene: most pixels stay undrawn, and every tile with a splat has some of both kinds
    d = raster_inputs(n, W, H, bw, {}, scale_lo=0.003, scale_hi=0.03)
    depths = np.random.default_rng(3).uniform(1, 5, n).astype(np.float32)
    ids, bins = cu(d["vs"]), cu(d["bins"])
    args = (cu(d["xys"]), cu(d["conics"]), cu(d["colors"]))
    opac, bg = cu(d["opac"]), cu(d["bg"])
    if rgbd:
        f = C.rasterize_forward_rgbd(d["tb"], (W, H, 1), ids, bins, *args, cu(depths), opac, bg, 0.0)
        Ts, idx = f[2], f[3]
    else:
        f = C.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), ids, bins, *args, opac, bg)
        Ts, idx = f[1], f[2]
    undrawn = Ts == 1.0
    assert 0.2 < undrawn.float().mean().item() < 0.98
    g = torch.Generator(device=DEV).manual_seed(4)
    v_img = torch.rand(H, W, 3, device=DEV, generator=g) * 2 - 1
    v_alpha = torch.rand(H, W, device=DEV, generator=g) * 2 - 1
    v_ext = torch.rand(H, W, device=DEV, generator=g) * 2 - 1

    def run(fill):
        vi, va, ve = v_img.clone(), v_alpha.clone(), v_ext.clone()
        vi[undrawn], va[undrawn], ve[undrawn] = fill, fill, fill
        if rgbd:
            return C.rasterize_backward_rgbd(H, W, ids, bins, *args, cu(depths), opac, bg, 0.0, Ts, idx, vi, ve, va)
        return C.rasterize_backward(H, W, bw, ids, bins, *args, opac, bg, Ts, idx, vi, va)

    clean, poisoned = run(0.0), run(float("nan"))
    for a, b in zip(clean, poisoned):
        assert torch.isfinite(b).all()
        scale = a.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale  # equal up to the order of the float atomics
    assert sum(float(a.abs().sum()) for a in clean) > 0


@pytest.mark.parametrize("n,W,H,ck,kw", [(10_000, 256, 256, {}, dict(scale_lo=0.005, scale_hi=0.05)),
                                         (4_000, 200, 120, dict(yaw=0.2, pitch=-0.1), dict(scale_lo=0.02, scale_hi=0.3)),
                                         (20_000, 97, 61, {}, dict(scale_lo=0.01, scale_hi=0.1))])
def test_scan_mapping_forward_equals_the_serial_walk(n, W, H, ck, kw):
    """north_star's "wave-64 prefix-scan for per-pixel transmittance" mapping (gsr_rasterize_forward_scan:
    lanes over splats, DPP prefix product, ballot termination) against the oracle and the serial tile16
    kernel: image / T 1e-4 abs on decision-stable pixels, final_idx exact there."""
    import rasterizer.cuda as C

    bw = 16
    d = raster_inputs(n, W, H, bw, ck, **kw)
    ref = O.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), d["vs"], d["bins"], d["xys"], d["conics"],
                              d["colors"], d["opac"], d["bg"], ambig_eps=1e-5)
    args = (cu(d["vs"]), cu(d["bins"]), cu(d["xys"]), cu(d["conics"]), cu(d["colors"]), cu(d["opac"]), cu(d["bg"]))
    out, Ts, idx = C.rasterize_forward_scan(d["tb"], (W, H, 1), *args)
    check_image(npy(out), npy(Ts), ref, ref[3])
    ok = ~ref[3]
    assert np.array_equal(npy(idx)[ok], ref[2][ok])
    out2, Ts2, idx2 = C.rasterize_forward(d["tb"], (bw, bw, 1), (W, H, 1), *args)
    assert (out - out2).abs()[cu(ok)].max().item() < 2e-6 and (Ts - Ts2).abs()[cu(ok)].max().item() < 2e-6


@pytest.mark.parametrize("n,W,H,lo,hi,frac,rgbd", [
    (60_000, 320, 208, 0.02, 0.12, 0.15, False),    # dense: every tile saturates inside the prefix lists
    (60_000, 320, 208, 0.02, 0.12, 0.15, True),
    (20_000, 317, 203, 0.004, 0.03, 0.3, False),    # sparse: background shows, most tiles stay unfinished
    (150_000, 640, 360, 0.01, 0.06, 0.05, True),    # a short prefix: many tiles need the second round
    (3_000, 160, 96, 0.01, 0.1, 0.5, False),
])
def test_two_round_lists_equal_the_single_walk(n, W, H, lo, hi, frac, rgbd, monkeypatch):
    """Prefix lists + saturation filter + second-round lists + resumed compositing (include/gsraster.h "two-round
    lists") against ONE walk over the full lists: image, final T (and the depth channel) bit-identical, the last
    contributing Gaussian of every drawn pixel the same, gradients equal up to the order of the float atomics --
    whatever the prefix length.  (The single walk without depth segments: the two rounds resume ONE chain.)"""
    import rasterizer.cuda as C

    monkeypatch.setattr(C, "depth_segments", lambda entries, num_tiles: (1, 0))
    cam = S.make_camera(W, H)
    sc = S.make_scene(n, cam, sh_degree=0, seed=11, scale_lo=lo, scale_hi=hi)
    cov3d, xys, depths, radii, conics, comp, tiles = (cu(a) for a in project_cpu(cam, sc, 16))
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    nt = tb[0] * tb[1]
    opac = cu(sc["opacities"])
    rng = np.random.default_rng(5)
    colors = cu(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    bg = cu(np.array(S.BACKGROUND, np.float32))
    v_img = cu(rng.uniform(-1, 1, (H, W, 3)).astype(np.float32))
    v_alpha = cu(rng.uniform(-1, 1, (H, W)).astype(np.float32))
    v_ext = cu(rng.uniform(-1, 1, (H, W)).astype(np.float32))
    extra = depths if rgbd else None

    # ---- the single walk over the full (exact) lists
    counts, recs = C.count_reach(xys, radii, conics, opac, tb)
    order, cum = C.depth_order(depths, radii, counts)
    I = int(cum[-1].item())
    ids, bins = C.bin_sorted(n, I, order, cum, xys, radii, tb, 16, recs)
    if rgbd:
        f = C.rasterize_forward_rgbd(tb, (W, H, 1), ids, bins, xys, conics, colors, depths, opac, bg, 0.0)
        img0, ext0, T0, idx0 = f[0], f[1], f[2], f[3]
        g0 = C.rasterize_backward_rgbd(H, W, ids, bins, xys, conics, colors, depths, opac, bg, 0.0, T0, idx0, v_img, v_ext,
                                       v_alpha)
    else:
        img0, T0, idx0 = C.rasterize_forward(tb, (16, 16, 1), (W, H, 1), ids, bins, xys, conics, colors, opac, bg)
        ext0 = None
        g0 = C.rasterize_backward(H, W, 16, ids, bins, xys, conics, colors, opac, bg, T0, idx0, v_img, v_alpha)

    # ---- two rounds
    _, recs2 = C.count_reach(xys, radii, conics, opac, tb, counts=False, extra_rows=1)
    n_culled = int((radii <= 0).sum())            # culled Gaussians sit at the FRONT of the depth order (key 0)
    n1 = n_culled + max(1, int(frac * (n - n_culled)))
    cap1, cap2 = I + 16, I + 16
    both = torch.empty(cap1 + cap2, dtype=torch.int32, device=DEV)
    c1, c2 = (torch.zeros(1, dtype=torch.int32, device=DEV) for _ in range(2))
    bins1 = C.tile_lists_subrange(order[:n1], cap1, recs2, tb, both[:cap1], c1)
    flags = torch.zeros(nt, dtype=torch.int32, device=DEV)
    img = torch.empty(H, W, 3, device=DEV)
    ext = torch.empty(H, W, device=DEV) if rgbd else None
    Ts = torch.empty(H, W, device=DEV)
    idx = torch.empty(H, W, dtype=torch.int32, device=DEV)
    C.rasterize_forward_round(1, tb, (W, H, 1), both, bins1, 0, xys, conics, colors, extra, opac, bg, 0.0, img, ext, Ts, idx,
                              flags)
    stats = torch.zeros(2, dtype=torch.int32, device=DEV)
    order2 = C.saturation_filter(order[n1:], recs2, n, flags, tb, stats)
    bins2 = C.tile_lists_subrange(order2, cap2, recs2, tb, both[cap1:], c2)
    C.rasterize_forward_round(2, tb, (W, H, 1), both, bins2, cap1, xys, conics, colors, extra, opac, bg, 0.0, img, ext, Ts, idx,
                              flags)
    torch.cuda.synchronize()
    k1, k2 = int(c1[0]), int(c2[0])
    assert 0 < k1 <= I and k1 + k2 <= I, (k1, k2, I)   # the filter only ever drops entries
    assert int(stats[0]) == int((flags != 0).sum()) and 0 <= int(stats[1]) <= n - n1
    assert torch.equal(img, img0) and torch.equal(Ts, T0)
    if rgbd:
        assert torch.equal(ext, ext0)
    drawn = T0 < 1.0
    assert torch.equal(both[idx.long()][drawn], ids[idx0.long()][drawn])
    g = C.rasterize_backward_two(H, W, both, bins1, bins2, cap1, xys, conics, colors, extra, opac, bg, 0.0, Ts, idx, v_img,
                                 v_ext if rgbd else None, v_alpha)
    assert len(g) == len(g0)
    for a, b in zip(g0, g):
        scale = a.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-12
    # a dense scene: far fewer entries are built than the full lists hold
    if frac == 0.15:
        assert k1 + k2 < 0.6 * I, (k1, k2, I)
    print(f"two rounds: {k1} + {k2} of {I} entries built, {int(stats[0])} of {nt} tiles unfinished after round 1")


@pytest.mark.parametrize("degree,deg_use", [(0, 0), (1, 0), (1, 1), (2, 2), (3, 1), (3, 3)])
@pytest.mark.parametrize("views", [1, 3, 8])
def test_sh_backward_over_several_views(degree, deg_use, views):
    """gsr_sh_backward_views (what data-parallel ranks run on their gathered colour cotangents) == scale x the sum over
    the views of the single-view SH backward, in both output layouts, views read straight out of one gathered
    [V, 3 N + 3] message (strided rows), tail of a workgroup included."""
    import rasterizer.cuda as C
    from gs_fused import sh_backward_views

    n = 10_007
    rng = np.random.default_rng(10 * degree + views)
    means = torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)).cuda()
    msg = torch.from_numpy(rng.standard_normal((views, 3 * n + 3)).astype(np.float32)).cuda()
    msg[:, 3 * n:] = torch.from_numpy(rng.uniform(4, 6, (views, 3)).astype(np.float32)).cuda()
    v_all, campos = msg[:, :3 * n], msg[:, 3 * n:]
    K = (degree + 1) ** 2
    ref = torch.zeros((n, K, 3), dtype=torch.float64, device="cuda")
    for r in range(views):
        d = means - campos[r]
        d = d / d.norm(dim=-1, keepdim=True)
        ref += C.compute_sh_backward(n, degree, deg_use, d.contiguous(), v_all[r].reshape(n, 3).contiguous()).double()
    ref *= 1.0 / views
    tol = 2e-6 * float(ref.abs().max()) + 1e-7
    joint = sh_backward_views(degree, deg_use, means, campos, v_all, 1.0 / views, split=False)
    assert joint.shape == (n, K, 3) and float((joint.double() - ref).abs().max()) <= tol
    v_dc, v_rest = sh_backward_views(degree, deg_use, means, campos, v_all, 1.0 / views, split=True)
    assert v_rest.shape == (n, K - 1, 3)
    assert torch.equal(v_dc, joint[:, 0, :]) and torch.equal(v_rest, joint[:, 1:, :])
    if deg_use < degree:
        assert not joint[:, (deg_use + 1) ** 2:, :].any()  # bands above the warm-up degree: exact zeros
