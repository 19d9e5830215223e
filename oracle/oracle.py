"""numpy/ctypes front end of the CPU oracle (``gsr_oracle.c``).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker -- never as the thing measured or shipped.  The product package
(``gaussian-splatting-toolkit_amd/rasterizer``) does not import it and fails
loudly when its HIP library is missing.

Function names/argument order follow the reference's native module
(``rasterizer/cuda/csrc/ext.cpp:6-17``, ``bindings.h:19-115``) so parity tests
read like calls into the reference.  All arrays are numpy, C-contiguous.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile ``libgsr_oracle.so`` with gcc (seconds)."""
    src = os.path.join(_HERE, "gsr_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libgsr_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gsr_oracle_cumsum.restype = C.c_int
        _lib.gsr_oracle_num_threads.restype = C.c_int
    return _lib


def num_threads() -> int:
    return int(lib().gsr_oracle_num_threads())


def set_threads(n: int) -> None:
    lib().gsr_oracle_set_threads(C.c_int(int(n)))


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def num_sh_bases(degree: int) -> int:
    return {0: 1, 1: 4, 2: 9, 3: 16}.get(degree, 25)


def project_gaussians_forward(
    num_points, means3d, scales, glob_scale, quats, viewmat, projmat,
    fx, fy, cx, cy, img_height, img_width, block_width, clip_thresh=0.01,
):
    """-> (cov3d, xys, depths, radii, conics, compensation, num_tiles_hit)
    (native tuple order, bindings.cu:158)."""
    n = int(num_points)
    means3d, scales, quats = _f(means3d), _f(scales), _f(quats)
    viewmat, projmat = _f(viewmat).reshape(-1), _f(projmat).reshape(-1)
    assert viewmat.size >= 12 and projmat.size == 16
    cov3d = np.empty((n, 6), np.float32)
    xys = np.empty((n, 2), np.float32)
    depths = np.empty((n,), np.float32)
    radii = np.empty((n,), np.int32)
    conics = np.empty((n, 3), np.float32)
    comp = np.empty((n,), np.float32)
    tiles = np.empty((n,), np.int32)
    lib().gsr_oracle_project_forward(
        C.c_int(n), _p(means3d), _p(scales), C.c_float(glob_scale), _p(quats),
        _p(viewmat), _p(projmat), C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_int(img_height), C.c_int(img_width),
        C.c_int(block_width), C.c_float(clip_thresh), _p(cov3d), _p(xys),
        _p(depths), _p(radii), _p(conics), _p(comp), _p(tiles),
    )
    return cov3d, xys, depths, radii, conics, comp, tiles


def project_gaussians_backward(
    num_points, means3d, scales, glob_scale, quats, viewmat, projmat,
    fx, fy, cx, cy, img_height, img_width, cov3d, radii, conics, compensation,
    v_xy, v_depth, v_conic, v_compensation,
):
    """-> (v_cov2d, v_cov3d, v_mean3d, v_scale, v_quat)"""
    n = int(num_points)
    args = [_f(means3d), _f(scales)]
    quats = _f(quats)
    viewmat, projmat = _f(viewmat).reshape(-1), _f(projmat).reshape(-1)
    cov3d, radii, conics, compensation = _f(cov3d), _i(radii), _f(conics), _f(compensation)
    v_xy, v_depth, v_conic, v_compensation = _f(v_xy), _f(v_depth), _f(v_conic), _f(v_compensation)
    v_cov2d = np.empty((n, 3), np.float32)
    v_cov3d = np.empty((n, 6), np.float32)
    v_mean3d = np.empty((n, 3), np.float32)
    v_scale = np.empty((n, 3), np.float32)
    v_quat = np.empty((n, 4), np.float32)
    lib().gsr_oracle_project_backward(
        C.c_int(n), _p(args[0]), _p(args[1]), C.c_float(glob_scale), _p(quats),
        _p(viewmat), _p(projmat), C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_int(img_height), C.c_int(img_width), _p(cov3d),
        _p(radii), _p(conics), _p(compensation), _p(v_xy), _p(v_depth),
        _p(v_conic), _p(v_compensation), _p(v_cov2d), _p(v_cov3d), _p(v_mean3d),
        _p(v_scale), _p(v_quat),
    )
    return v_cov2d, v_cov3d, v_mean3d, v_scale, v_quat


def compute_sh_forward(num_points, degree, degrees_to_use, viewdirs, coeffs):
    n = int(num_points)
    viewdirs, coeffs = _f(viewdirs), _f(coeffs)
    assert coeffs.shape == (n, num_sh_bases(degree), 3)
    colors = np.empty((n, 3), np.float32)
    lib().gsr_oracle_sh_forward(
        C.c_int(n), C.c_int(degree), C.c_int(degrees_to_use), _p(viewdirs),
        _p(coeffs), _p(colors),
    )
    return colors


def compute_sh_backward(num_points, degree, degrees_to_use, viewdirs, v_colors):
    n = int(num_points)
    viewdirs, v_colors = _f(viewdirs), _f(v_colors)
    v_coeffs = np.empty((n, num_sh_bases(degree), 3), np.float32)
    lib().gsr_oracle_sh_backward(
        C.c_int(n), C.c_int(degree), C.c_int(degrees_to_use), _p(viewdirs),
        _p(v_colors), _p(v_coeffs),
    )
    return v_coeffs


def compute_cumulative_intersects(num_tiles_hit) -> Tuple[int, np.ndarray]:
    t = _i(num_tiles_hit)
    cum = np.empty_like(t)
    total = lib().gsr_oracle_cumsum(C.c_int(t.size), _p(t), _p(cum))
    return int(total), cum


def map_gaussian_to_intersects(
    num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds, block_width
):
    xys, depths, radii, cum = _f(xys), _f(depths), _i(radii), _i(cum_tiles_hit)
    isect = np.zeros((int(num_intersects),), np.int64)
    gids = np.zeros((int(num_intersects),), np.int32)
    lib().gsr_oracle_map_intersects(
        C.c_int(int(num_points)), _p(xys), _p(depths), _p(radii), _p(cum),
        C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), C.c_int(block_width),
        _p(isect), _p(gids),
    )
    return isect, gids


def sort_intersects(isect_ids, gaussian_ids):
    isect_ids = np.ascontiguousarray(isect_ids, np.int64)
    gaussian_ids = _i(gaussian_ids)
    ks = np.empty_like(isect_ids)
    vs = np.empty_like(gaussian_ids)
    lib().gsr_oracle_sort_intersects(
        C.c_int(isect_ids.size), _p(isect_ids), _p(gaussian_ids), _p(ks), _p(vs)
    )
    return ks, vs


def get_tile_bin_edges(num_intersects, isect_ids_sorted, tile_bounds):
    ks = np.ascontiguousarray(isect_ids_sorted, np.int64)
    nt = int(tile_bounds[0]) * int(tile_bounds[1])
    bins = np.empty((nt, 2), np.int32)
    lib().gsr_oracle_tile_bin_edges(C.c_int(int(num_intersects)), _p(ks), C.c_int(nt), _p(bins))
    return bins


def bin_and_sort_gaussians(
    num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds, block_width
):
    isect, gids = map_gaussian_to_intersects(
        num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds, block_width
    )
    ks, vs = sort_intersects(isect, gids)
    bins = get_tile_bin_edges(num_intersects, ks, tile_bounds)
    return isect, gids, ks, vs, bins


def rasterize_forward(
    tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics,
    colors, opacities, background, ambig_eps: Optional[float] = None,
):
    """-> (out_img [H,W,C], final_Ts, final_idx[, ambig mask])"""
    W, H = int(img_size[0]), int(img_size[1])
    colors = _f(colors)
    ch = colors.shape[1]
    gids, bins = _i(gaussian_ids_sorted), _i(tile_bins)
    xys, conics, opac, bg = _f(xys), _f(conics), _f(opacities).reshape(-1), _f(background)
    out = np.empty((H, W, ch), np.float32)
    Ts = np.empty((H, W), np.float32)
    idx = np.empty((H, W), np.int32)
    amb = np.zeros((H, W), np.uint8) if ambig_eps is not None else None
    lib().gsr_oracle_rasterize_forward(
        C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), C.c_int(block[0]),
        C.c_int(W), C.c_int(H), C.c_int(ch), _p(gids), _p(bins), _p(xys),
        _p(conics), _p(colors), _p(opac), _p(bg), _p(out), _p(Ts), _p(idx),
        _p(amb) if amb is not None else None,
        C.c_float(ambig_eps if ambig_eps is not None else 0.0),
    )
    if amb is not None:
        return out, Ts, idx, amb.astype(bool)
    return out, Ts, idx


def rasterize_backward(
    img_height, img_width, block_width, gaussian_ids_sorted, tile_bins, xys,
    conics, colors, opacities, background, final_Ts, final_idx, v_output,
    v_output_alpha, with_abs_sums: bool = False, ambig_eps: Optional[float] = None,
):
    """-> (v_xy, v_conic, v_colors, v_opacity[N,1]) [+ the same four shapes holding
    the sum of |per-pixel terms|, the scale fp32 accumulation error lives on]"""
    colors = _f(colors)
    n, ch = colors.shape
    gids, bins = _i(gaussian_ids_sorted), _i(tile_bins)
    xys, conics, opac, bg = _f(xys), _f(conics), _f(opacities).reshape(-1), _f(background)
    Ts, fidx = _f(final_Ts), _i(final_idx)
    vo, voa = _f(v_output), _f(v_output_alpha)
    v_xy = np.empty((n, 2), np.float32)
    v_conic = np.empty((n, 3), np.float32)
    v_colors = np.empty((n, ch), np.float32)
    v_opac = np.empty((n, 1), np.float32)
    abs_sums = np.empty((n, 6 + ch), np.float32) if with_abs_sums else None
    amb = np.zeros((n,), np.uint8) if ambig_eps is not None else None
    lib().gsr_oracle_rasterize_backward(
        C.c_int(img_height), C.c_int(img_width), C.c_int(block_width), C.c_int(ch),
        C.c_int(n), _p(gids), _p(bins), _p(xys), _p(conics), _p(colors), _p(opac),
        _p(bg), _p(Ts), _p(fidx), _p(vo), _p(voa), _p(v_xy), _p(v_conic),
        _p(v_colors), _p(v_opac), _p(abs_sums) if abs_sums is not None else None,
        _p(amb) if amb is not None else None, C.c_float(ambig_eps if ambig_eps is not None else 0.0),
    )
    out = (v_xy, v_conic, v_colors, v_opac)
    if with_abs_sums:
        out = out + (abs_sums[:, 0:2], abs_sums[:, 2:5], abs_sums[:, 6:], abs_sums[:, 5:6])
    if amb is not None:
        out = out + (amb.astype(bool),)
    return out


# generic-channel entry points share the implementation (fp32 accumulators)
nd_rasterize_forward = rasterize_forward
nd_rasterize_backward = rasterize_backward


def compute_cov2d_bounds(num_pts, cov2d):
    cov2d = _f(cov2d)
    conics = np.empty((int(num_pts), 3), np.float32)
    radii = np.empty((int(num_pts), 1), np.float32)
    lib().gsr_oracle_cov2d_bounds(C.c_int(int(num_pts)), _p(cov2d), _p(conics), _p(radii))
    return conics, radii


# ---------------------------------------------------------------- pipelines


def render_forward(
    means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
    block_width, colors, opacities, background, clip_thresh=0.01, ambig_eps=None,
):
    """project -> bin/sort -> composite, the order `_RasterizeGaussians.forward`
    (rasterize.py:93-183) runs them in.  Returns a dict of every intermediate."""
    n = means3d.shape[0]
    cov3d, xys, depths, radii, conics, comp, tiles = project_gaussians_forward(
        n, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
        H, W, block_width, clip_thresh,
    )
    tb = ((W + block_width - 1) // block_width, (H + block_width - 1) // block_width, 1)
    I, cum = compute_cumulative_intersects(tiles)
    res = dict(cov3d=cov3d, xys=xys, depths=depths, radii=radii, conics=conics,
               compensation=comp, num_tiles_hit=tiles, num_intersects=I,
               cum_tiles_hit=cum, tile_bounds=tb)
    if I < 1:
        return res
    isect, gids, ks, vs, bins = bin_and_sort_gaussians(n, I, xys, depths, radii, cum, tb, block_width)
    r = rasterize_forward(tb, (block_width, block_width, 1), (W, H, 1), vs, bins, xys,
                          conics, colors, opacities, background, ambig_eps=ambig_eps)
    res.update(isect_ids=isect, gaussian_ids=gids, isect_ids_sorted=ks,
               gaussian_ids_sorted=vs, tile_bins=bins, out_img=r[0], final_Ts=r[1],
               final_idx=r[2])
    if ambig_eps is not None:
        res["ambig"] = r[3]
    return res


def l1_ssim_loss(pred, gt, ssim_lambda=0.2, with_grad=True):
    """(1-lambda)*L1 + lambda*(1-SSIM) of two [H,W,3] images and d loss / d pred
    (vanilla_gs.py:926-944 + pytorch_msssim 1.0.0's SSIM).  -> (loss, l1, ssim, v_pred)"""
    pred, gt = _f(pred), _f(gt)
    H, W, _ = pred.shape
    l1, ss = C.c_double(), C.c_double()
    v = np.empty_like(pred) if with_grad else None
    fn = lib().gsr_oracle_l1_ssim
    fn.restype = C.c_double
    loss = fn(C.c_int(H), C.c_int(W), _p(pred), _p(gt), C.c_float(ssim_lambda), C.byref(l1), C.byref(ss),
              _p(v) if v is not None else None)
    return float(loss), l1.value, ss.value, v


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One step of torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False)
    as `_single_tensor_adam` (torch/optim/adam.py) computes it for fp32 tensors:
    fp32 elementwise arithmetic, bias corrections as Python floats.  `step` counts
    from 1.  -> (param, exp_avg, exp_avg_sq), new arrays.  This is what the
    toolkit's per-group optimisers do (gs_toolkit/engine/optimizers.py:59-196)."""
    f = np.float32
    p, g = np.asarray(param, f), np.asarray(grad, f)
    m = (np.asarray(exp_avg, f) * f(beta1) + g * f(1.0 - beta1)).astype(f)
    v = (np.asarray(exp_avg_sq, f) * f(beta2) + (g * g) * f(1.0 - beta2)).astype(f)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = (np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)).astype(f)
    p = (p - f(step_size) * (m / denom)).astype(f)
    return p, m, v
