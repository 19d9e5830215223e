"""``rasterizer.cuda`` -- the native module of the rasterizer, MI355X edition.

The reference exposes an 11-function pybind module under this name
(``rasterizer/cuda/csrc/ext.cpp:6-17``; every Python module of the package does
``import rasterizer.cuda as _C``).  The module name is kept so that those
imports keep working; the implementation is hand-written HIP for gfx950 behind
the C ABI of ``include/gsraster.h``, bound here with ctypes.  Each function has
the reference's name, argument order, returned tuple order, dtypes, shapes and
error type (``RuntimeError`` for what ``TORCH_CHECK``/``AT_ERROR`` raise there).

PyTorch is used for device memory and streams only: outputs are allocated with
``torch.empty`` on the inputs' device (the kernels write every element, so the
reference's ``torch::zeros`` pre-fill is not needed) and kernels are enqueued
on the *current* HIP stream of that device.  There is no CPU path.
"""
import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _tuning
from ._backend import lib as _lib

_f32, _i32, _i64 = torch.float32, torch.int32, torch.int64


def _check(t: Tensor, name: str, dtype=None) -> Tensor:
    # CHECK_INPUT of the reference (bindings.h:10-15)
    if not isinstance(t, Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t


def _ptr(t: Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


class _Already:
    """(no-op context: the tensors' device is the current one already)"""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_ALREADY = _Already()


def _on(dev):
    """`torch.cuda.device(dev)` -- allocations and launches go to the inputs' device -- without the context
    manager's ~3 us when that device is current already (a dozen wrappers per view)."""
    idx = dev.index
    return _ALREADY if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(dev)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device) -> C.c_void_p:
    """The device's current stream as the C ABI takes it (the raw handle where torch hands it out directly: the
    `torch.cuda.Stream` object of `current_stream()` costs ~5 us, and every wrapper asks)."""
    idx = device.index
    if _raw_stream is not None and idx is not None:
        return C.c_void_p(_raw_stream(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _call(fn_name: str, *args) -> None:
    rc = getattr(_lib(), fn_name)(*args)
    if rc != 0:
        raise RuntimeError(f"{fn_name} failed ({rc}): {_lib().gsr_last_error().decode()}")


def _cf(v) -> C.c_float:
    return C.c_float(float(v))


def project_gaussians_forward(
    num_points: int, means3d: Tensor, scales: Optional[Tensor], glob_scale: float, quats: Optional[Tensor],
    viewmat: Tensor, projmat: Tensor, fx: float, fy: float, cx: float, cy: float,
    img_height: int, img_width: int, block_width: int, clip_thresh: float,
    cov3d_precomp: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (cov3d, xys, depths, radii, conics, compensation, num_tiles_hit);
    replaces ``project_gaussians_forward_tensor`` (bindings.cu:107-160).
    ``cov3d_precomp`` [N,6] (with ``scales`` = ``quats`` = None): covariances handed in
    instead of being built from scales and rotations; returned as ``cov3d``."""
    precomp = cov3d_precomp is not None
    if precomp != (scales is None) or precomp != (quats is None):
        raise RuntimeError("pass either scales and quats, or cov3d_precomp")
    for t, nm in ((means3d, "means3d"), (viewmat, "viewmat"), (projmat, "projmat")) + (
            ((cov3d_precomp, "cov3d_precomp"),) if precomp else ((scales, "scales"), (quats, "quats"))):
        _check(t, nm, _f32)
    n = int(num_points)
    if viewmat.numel() < 12 or projmat.numel() != 16:
        raise RuntimeError("viewmat must hold at least 3x4 and projmat 4x4 values")
    if means3d.numel() != 3 * n or (not precomp and (scales.numel() != 3 * n or quats.numel() != 4 * n)) \
            or (precomp and cov3d_precomp.numel() != 6 * n):
        raise RuntimeError("means3d/scales/quats do not match num_points")
    dev = means3d.device
    _opt0 = lambda t: None if t is None else _ptr(t)
    with _on(dev):
        # seven outputs, seven allocations (the caching allocator hands each out in ~1 us): in round 4 they were carved
        # out of ONE allocation, which made them views sharing a storage and a version counter (a caller's
        # in-place op on one invalidated the others that autograd had saved, and keeping `radii` alive pinned all 15
        # words per Gaussian); independent tensors over a shared storage (`set_`) cost 3 us apiece -- more than this
        e = lambda shape, dt=_f32: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
        xys, depths, conics, compensation = e((n, 2)), e((n,)), e((n, 3)), e((n,))
        radii, num_tiles_hit = e((n,), _i32), e((n,), _i32)
        cov3d = cov3d_precomp if precomp else e((n, 6))
        _call(
            "gsr_project_forward", C.c_int(n), _ptr(means3d), _opt0(scales), _cf(glob_scale),
            _opt0(quats), _ptr(viewmat), _ptr(projmat), _cf(fx), _cf(fy), _cf(cx), _cf(cy),
            C.c_uint(img_height), C.c_uint(img_width), C.c_uint(block_width), _cf(clip_thresh),
            _ptr(cov3d), _ptr(xys), _ptr(depths), _ptr(radii), _ptr(conics), _ptr(compensation),
            _ptr(num_tiles_hit), _stream(dev),
        )
    return cov3d, xys, depths, radii, conics, compensation, num_tiles_hit


def project_gaussians_backward(
    num_points: int, means3d: Tensor, scales: Tensor, glob_scale: float, quats: Tensor,
    viewmat: Tensor, projmat: Tensor, fx: float, fy: float, cx: float, cy: float,
    img_height: int, img_width: int, cov3d: Tensor, radii: Tensor, conics: Tensor,
    compensation: Tensor, v_xy: Tensor, v_depth: Tensor, v_conic: Tensor, v_compensation: Tensor,
    trusted: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (v_cov2d, v_cov3d, v_mean3d, v_scale, v_quat);
    replaces ``project_gaussians_backward_tensor`` (bindings.cu:164-216).  ``trusted``: the first sixteen arguments
    are the tensors a forward call of this module validated and produced (what the autograd node passes back): only
    the cotangents are checked."""
    n = int(num_points)
    dev = means3d.device
    precomp = scales is None and quats is None  # covariances were handed in: the chain ends at v_cov3d
    if not trusted:
        for t, nm in ((means3d, "means3d"), (viewmat, "viewmat"),
                      (projmat, "projmat"), (cov3d, "cov3d"), (conics, "conics"),
                      (compensation, "compensation")) + (() if precomp else ((scales, "scales"), (quats, "quats"))):
            _check(t, nm, _f32)
        _check(radii, "radii", _i32)
    # cotangents may arrive non-contiguous / expanded from autograd; None = zero
    v_xy, v_depth, v_conic, v_compensation = (
        None if t is None else _check(t.contiguous(), nm, _f32)
        for t, nm in ((v_xy, "v_xy"), (v_depth, "v_depth"), (v_conic, "v_conic"),
                      (v_compensation, "v_compensation"))
    )
    _opt = lambda t: None if t is None else _ptr(t)
    with _on(dev):
        e = lambda shape: torch.empty(shape, dtype=_f32, device=dev)  # noqa: E731
        v_cov2d, v_cov3d, v_mean3d = e((n, 3)), e((n, 6)), e((n, 3))
        v_scale = None if precomp else e((n, 3))
        v_quat = None if precomp else e((n, 4))
        _call(
            "gsr_project_backward", C.c_int(n), _ptr(means3d), _opt(scales), _cf(glob_scale),
            _opt(quats), _ptr(viewmat), _ptr(projmat), _cf(fx), _cf(fy), _cf(cx), _cf(cy),
            C.c_uint(img_height), C.c_uint(img_width), _ptr(cov3d), _ptr(radii), _ptr(conics),
            _ptr(compensation), _opt(v_xy), _opt(v_depth), _opt(v_conic), _opt(v_compensation),
            _ptr(v_cov2d), _ptr(v_cov3d), _ptr(v_mean3d), _opt(v_scale), _opt(v_quat),
            _stream(dev),
        )
    return v_cov2d, v_cov3d, v_mean3d, v_scale, v_quat


def _num_sh_bases(degree: int) -> int:
    return {0: 1, 1: 4, 2: 9, 3: 16}.get(int(degree), 25)


def compute_sh_forward(num_points: int, degree: int, degrees_to_use: int, viewdirs: Tensor,
                       coeffs: Tensor) -> Tensor:
    """-> colors [N,3]; replaces ``compute_sh_forward_tensor`` (bindings.cu:58-77)."""
    n = int(num_points)
    if coeffs.dim() != 3 or coeffs.size(0) != n or coeffs.size(1) != _num_sh_bases(degree) \
            or coeffs.size(2) != 3:
        raise RuntimeError("coeffs must have dimensions (N, D, 3)")
    _check(viewdirs, "viewdirs", _f32)
    _check(coeffs, "coeffs", _f32)
    dev = coeffs.device
    with _on(dev):
        colors = torch.empty((n, 3), dtype=_f32, device=dev)
        _call("gsr_sh_forward", C.c_uint(n), C.c_uint(degree), C.c_uint(degrees_to_use),
              _ptr(viewdirs), _ptr(coeffs), _ptr(colors), _stream(dev))
    return colors


def compute_sh_backward(num_points: int, degree: int, degrees_to_use: int, viewdirs: Tensor,
                        v_colors: Tensor, trusted: bool = False) -> Tensor:
    """-> v_coeffs [N,K,3]; replaces ``compute_sh_backward_tensor`` (bindings.cu:79-103)."""
    n = int(num_points)
    if viewdirs.dim() != 2 or viewdirs.size(0) != n or viewdirs.size(1) != 3:
        raise RuntimeError("viewdirs must have dimensions (N, 3)")
    if v_colors.dim() != 2 or v_colors.size(0) != n or v_colors.size(1) != 3:
        raise RuntimeError("v_colors must have dimensions (N, 3)")
    if not trusted:  # (trusted: `viewdirs` is the tensor the forward call validated)
        _check(viewdirs, "viewdirs", _f32)
    v_colors = _check(v_colors.contiguous(), "v_colors", _f32)
    dev = viewdirs.device
    with _on(dev):
        v_coeffs = torch.empty((n, _num_sh_bases(degree), 3), dtype=_f32, device=dev)
        _call("gsr_sh_backward", C.c_uint(n), C.c_uint(degree), C.c_uint(degrees_to_use),
              _ptr(viewdirs), _ptr(v_colors), _ptr(v_coeffs), _stream(dev))
    return v_coeffs


def cumsum_tiles(num_tiles_hit: Tensor) -> Tensor:
    """Inclusive int32 scan (not part of the reference's native module, which
    uses ``torch.cumsum``; exported for ``utils.compute_cumulative_intersects``)."""
    _check(num_tiles_hit, "num_tiles_hit", _i32)
    n = num_tiles_hit.numel()
    dev = num_tiles_hit.device
    with _on(dev):
        cum = torch.empty_like(num_tiles_hit)
        nbytes = int(_lib().gsr_cumsum_workspace_bytes(C.c_int(n)))
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        _call("gsr_cumsum_tiles", C.c_int(n), _ptr(num_tiles_hit), _ptr(cum), _ptr(ws),
              C.c_size_t(nbytes), _stream(dev))
    return cum


def map_gaussian_to_intersects(
    num_points: int, num_intersects: int, xys: Tensor, depths: Tensor, radii: Tensor,
    cum_tiles_hit: Tensor, tile_bounds: Tuple[int, int, int], block_width: int,
) -> Tuple[Tensor, Tensor]:
    """-> (isect_ids i64[I], gaussian_ids i32[I]);
    replaces ``map_gaussian_to_intersects_tensor`` (bindings.cu:218-251)."""
    _check(xys, "xys", _f32)
    _check(depths, "depths", _f32)
    _check(radii, "radii", _i32)
    _check(cum_tiles_hit, "cum_tiles_hit", _i32)
    dev = xys.device
    I = int(num_intersects)
    with _on(dev):
        # zero-filled like the reference: slots not claimed by any splat stay 0
        isect_ids = torch.zeros((I,), dtype=_i64, device=dev)
        gaussian_ids = torch.zeros((I,), dtype=_i32, device=dev)
        _call("gsr_map_intersects", C.c_int(int(num_points)), C.c_int(I), _ptr(xys), _ptr(depths),
              _ptr(radii), _ptr(cum_tiles_hit), C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]),
              C.c_uint(block_width), _ptr(isect_ids), _ptr(gaussian_ids), _stream(dev))
    return isect_ids, gaussian_ids


def sort_intersects(isect_ids: Tensor, gaussian_ids: Tensor, num_tiles: int) -> Tuple[Tensor, Tensor]:
    """Stable radix sort of the (tile|depth) keys with their Gaussian ids (the
    reference's ``torch.sort`` + ``torch.gather``, utils.py:179-180)."""
    _check(isect_ids, "isect_ids", _i64)
    _check(gaussian_ids, "gaussian_ids", _i32)
    I = isect_ids.numel()
    dev = isect_ids.device
    with _on(dev):
        keys = torch.empty_like(isect_ids)
        vals = torch.empty_like(gaussian_ids)
        nbytes = int(_lib().gsr_sort_workspace_bytes(C.c_int(I)))
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        _call("gsr_sort_intersects", C.c_int(I), C.c_int(max(int(num_tiles), 1)), _ptr(isect_ids),
              _ptr(gaussian_ids), _ptr(keys), _ptr(vals), _ptr(ws), C.c_size_t(nbytes), _stream(dev))
    return keys, vals


def get_tile_bin_edges(num_intersects: int, isect_ids_sorted: Tensor,
                       tile_bounds: Tuple[int, int, int]) -> Tensor:
    """-> tile_bins i32[T,2]; replaces ``get_tile_bin_edges_tensor`` (bindings.cu:253-267)."""
    _check(isect_ids_sorted, "isect_ids_sorted", _i64)
    dev = isect_ids_sorted.device
    nt = int(tile_bounds[0]) * int(tile_bounds[1])
    with _on(dev):
        tile_bins = alloc_tile_bins(tile_bounds, dev)
        _call("gsr_tile_bin_edges", C.c_int(int(num_intersects)), _ptr(isect_ids_sorted),
              C.c_int(nt), _ptr(tile_bins), _stream(dev))
    return tile_bins


def depth_order(depths: Tensor, radii: Tensor, num_tiles_hit: Optional[Tensor]) -> Tuple[Tensor, Optional[Tensor]]:
    """First half of the fused binning pipeline (``gsr_depth_order``):
    -> (order i32[N], cum_sorted i32[B*N]); ``cum_sorted[-1]`` is the number of
    intersections.  ``num_tiles_hit`` is [N], or the band-major [B*N] counts of
    :func:`count_reach` on a tile grid of B = :func:`tile_bands` bands, or None: the
    order only, ``cum_sorted`` is None (lists without counts, :func:`lists_need_counts`)."""
    _check(depths, "depths", _f32)
    _check(radii, "radii", _i32)
    n = depths.numel()
    bands = 1
    if num_tiles_hit is not None:
        _check(num_tiles_hit, "num_tiles_hit", _i32)
        bands = num_tiles_hit.numel() // n if n else 1
        if n and (num_tiles_hit.numel() != bands * n or bands < 1):
            raise RuntimeError("depth_order: num_tiles_hit must hold N (or bands * N) counts")
    dev = depths.device
    with _on(dev):
        order = torch.empty((n,), dtype=_i32, device=dev)
        cum = torch.empty((bands * n,), dtype=_i32, device=dev) if num_tiles_hit is not None else None
        nbytes = int(_lib().gsr_depth_order_workspace_bytes(C.c_int(n), C.c_int(bands)))
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        _call("gsr_depth_order", C.c_int(n), _ptr(depths), _ptr(radii),
              _ptr(num_tiles_hit) if num_tiles_hit is not None else None, C.c_int(bands),
              _ptr(order), _ptr(cum) if cum is not None else None, _ptr(ws), C.c_size_t(nbytes), _stream(dev))
    return order, cum


def lists_need_counts(num_points: int, num_intersects: int, tile_bounds: Tuple[int, int, int],
                      device_sized: bool = True, want_slots: bool = False) -> bool:
    """``gsr_bin_sorted_needs_counts``: False when :func:`bin_sorted` (exact lists, one
    band) will take the two-level partition for these sizes -- the caller may then skip
    the counts: ``count_reach(..., counts=False)``, ``depth_order(depths, radii, None)``,
    ``bin_sorted(..., cum_sorted=None)``."""
    return bool(_lib().gsr_bin_sorted_needs_counts(C.c_int(int(num_points)), C.c_int(int(num_intersects)),
                                                   C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]),
                                                   C.c_int(1 if device_sized else 0), C.c_int(1 if want_slots else 0)))


def tile_bands(tile_bounds: Tuple[int, int, int]) -> int:
    """``gsr_tile_bands``: 1 up to 16384 tiles; above, the number of tile-row bands the
    fused binning works in (4K at 16 px: 4)."""
    return int(_lib().gsr_tile_bands(C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1])))


def publish_int32(src: Tensor, dst: Tensor) -> None:
    """``gsr_publish_int32``: ``dst[0] = src[0]`` by a one-thread kernel in stream order;
    ``dst`` may be pinned host memory (int32)."""
    _check(src, "src", _i32)
    if dst.dtype != _i32 or dst.numel() < 1:
        raise RuntimeError("publish_int32: dst must be int32")
    dev = src.device
    with _on(dev):
        _call("gsr_publish_int32", _ptr(src), _ptr(dst), _stream(dev))


def count_reach(xys: Tensor, radii: Tensor, conics: Tensor, opacities: Tensor,
                tile_bounds: Tuple[int, int, int], bands: int = 1, counts: bool = True,
                extra_rows: int = 0) -> Tuple[Optional[Tensor], Tensor]:
    """``gsr_count_reach``: per Gaussian, the number of 16x16 tiles of its bounding
    box in which it can reach alpha >= 1/255 -> (counts i32[bands*N], band-major, summing to
    <= num_tiles_hit per Gaussian; opaque per-Gaussian records for :func:`bin_sorted`).
    ``bands``: 1 (default: one count per Gaussian; large grids then take the two-level
    partition) or :func:`tile_bands` (tile-row bands: needed for ``want_slots``).
    ``counts=False``: the records only (-> (None, records)), see :func:`lists_need_counts`."""
    _check(xys, "xys", _f32)
    _check(radii, "radii", _i32)
    _check(conics, "conics", _f32)
    _check(opacities, "opacities", _f32)
    n = radii.numel()
    if xys.numel() != 2 * n or conics.numel() != 3 * n or opacities.numel() != n:
        raise RuntimeError("count_reach: xys [N,2], conics [N,3], opacities [N,1] expected")
    dev = xys.device
    with _on(dev):
        cnt = torch.empty((int(bands) * n,), dtype=_i32, device=dev) if counts else None
        recs = torch.empty((n + int(extra_rows), int(_lib().gsr_reach_record_bytes())), dtype=torch.uint8, device=dev)
        if extra_rows:
            recs[n:].zero_()  # all-zero records are culled ones (empty tile box): `saturation_filter`'s dummy
        _call("gsr_count_reach", C.c_int(n), _ptr(xys), _ptr(radii), _ptr(conics), _ptr(opacities),
              C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), C.c_int(int(bands)),
              _ptr(cnt) if cnt is not None else None, _ptr(recs), _stream(dev))
    return cnt, recs


def reach_records_depth_order(xys: Tensor, radii: Tensor, conics: Tensor, opacities: Tensor, depths: Tensor,
                              tile_bounds: Tuple[int, int, int], extra_rows: int = 0) -> Tuple[Tensor, Tensor]:
    """``gsr_reach_records_depth_order``: ``count_reach(..., counts=False)`` and ``depth_order(depths, radii, None)``
    as one call -> (records, order), the same values (lists without counts, :func:`lists_need_counts`)."""
    _check(xys, "xys", _f32)
    _check(radii, "radii", _i32)
    _check(conics, "conics", _f32)
    _check(opacities, "opacities", _f32)
    _check(depths, "depths", _f32)
    n = radii.numel()
    if xys.numel() != 2 * n or conics.numel() != 3 * n or opacities.numel() != n or depths.numel() != n:
        raise RuntimeError("reach_records_depth_order: xys [N,2], conics [N,3], opacities [N,1], depths [N] expected")
    dev = xys.device
    with _on(dev):
        recs = torch.empty((n + int(extra_rows), int(_lib().gsr_reach_record_bytes())), dtype=torch.uint8, device=dev)
        if extra_rows:
            recs[n:].zero_()  # (as in count_reach: the dummy culled record)
        order = torch.empty((n,), dtype=_i32, device=dev)
        nbytes = int(_lib().gsr_depth_order_workspace_bytes(C.c_int(n), C.c_int(1)))
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        _call("gsr_reach_records_depth_order", C.c_int(n), _ptr(xys), _ptr(radii), _ptr(conics), _ptr(opacities),
              _ptr(depths), C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), _ptr(recs), _ptr(order), _ptr(ws),
              C.c_size_t(nbytes), _stream(dev))
    return recs, order


MAX_SCATTER_TILES = 16384  # tile grids the device-sized path supports without reach records (tile_scatter.hip)


def bin_sorted(num_points: int, num_intersects: int, order: Tensor, cum_sorted: Optional[Tensor], xys: Tensor,
               radii: Tensor, tile_bounds: Tuple[int, int, int], block_width: int,
               reach_records: Optional[Tensor] = None, device_sized: bool = False,
               count_out: Optional[Tensor] = None, want_slots: bool = False):
    """Second half (``gsr_bin_sorted``): -> (gaussian_ids_sorted i32[I],
    tile_bins i32[T,2]), identical to what ``bin_and_sort_gaussians`` returns.
    With the records (and counts) of :func:`count_reach`: the same lists without
    the pairs that cannot reach alpha >= 1/255 anywhere in their tile.
    ``device_sized``: ``num_intersects`` is only a capacity; the length is read on
    the device from ``cum_sorted[-1]`` and the lists are cut at the capacity
    (``gsr_bin_sorted_dev``) -- the caller checks ``cum_sorted[-1] <= capacity``
    later, off the critical path; ``count_out`` (int32[1], pinned host memory or
    device) receives that count.  ``cum_sorted=None``: only where :func:`lists_need_counts`
    is False.  ``want_slots``: also return ``slot_of_entry`` i32[I]
    (the inverse of the scatter, for :func:`rasterize_backward_det`)."""
    _check(order, "order", _i32)
    if cum_sorted is not None:
        _check(cum_sorted, "cum_sorted", _i32)
    _check(xys, "xys", _f32)
    _check(radii, "radii", _i32)
    if reach_records is not None:
        _check(reach_records, "reach_records", torch.uint8)
        if reach_records.numel() != int(num_points) * int(_lib().gsr_reach_record_bytes()):
            raise RuntimeError("bin_sorted: reach_records has the wrong size")
    I = int(num_intersects)
    nt = int(tile_bounds[0]) * int(tile_bounds[1])
    dev = xys.device
    with _on(dev):
        ids = torch.empty((I,), dtype=_i32, device=dev)
        tile_bins = alloc_tile_bins(tile_bounds, dev)
        nbytes = int(_lib().gsr_bin_sorted_workspace_bytes(C.c_int(int(num_points)), C.c_int(I),
                                                           C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1])))
        bands = cum_sorted.numel() // max(int(num_points), 1) if int(num_points) and cum_sorted is not None else 1
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        slots = torch.empty((I,), dtype=_i32, device=dev) if want_slots else None
        head = (C.c_int(int(num_points)), C.c_int(I), _ptr(order),
                _ptr(cum_sorted) if cum_sorted is not None else None, _ptr(xys), _ptr(radii),
                _ptr(reach_records) if reach_records is not None else None, C.c_int(tile_bounds[0]),
                C.c_int(tile_bounds[1]), C.c_uint(block_width), C.c_int(bands), _ptr(ids), _ptr(tile_bins))
        tail = (_ptr(slots) if want_slots else None, _ptr(ws), C.c_size_t(nbytes), _stream(dev))
        if device_sized:
            _call("gsr_bin_sorted_dev", *head, _ptr(count_out) if count_out is not None else None, *tail)
        else:
            _call("gsr_bin_sorted", *head, *tail)
    if want_slots:
        return ids, tile_bins, slots
    return ids, tile_bins


# ---- two-round lists (include/gsraster.h "two-round lists for deep scenes") -----------------------------
def tile_lists_subrange(order_sub: Tensor, capacity: int, reach_records: Tensor, tile_bounds, ids_out: Tensor,
                        count_out: Optional[Tensor] = None) -> Tensor:
    """``gsr_tile_lists_subrange``: the two-level partition over ``order_sub`` (a contiguous slice of the depth
    order) into ``ids_out`` (int32[capacity], e.g. a slice of the combined array) -> tile_bins i32[T,2] relative
    to ``ids_out``; ``count_out`` (int32[1], device or pinned) receives the uncut length."""
    _check(order_sub, "order", _i32)
    _check(ids_out, "ids_out", _i32)
    if ids_out.numel() < int(capacity):
        raise RuntimeError("tile_lists_subrange: ids_out is smaller than the capacity")
    n = order_sub.numel()
    nt = int(tile_bounds[0]) * int(tile_bounds[1])
    dev = ids_out.device
    with _on(dev):
        bins = alloc_tile_bins(tile_bounds, dev)
        nbytes = int(_lib().gsr_tile_lists_subrange_workspace_bytes(C.c_int(n), C.c_int(int(capacity)),
                                                                   C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1])))
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        _call("gsr_tile_lists_subrange", C.c_int(n), C.c_int(int(capacity)), _ptr(order_sub), _ptr(reach_records),
              C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), _ptr(ids_out), _ptr(bins),
              _ptr(count_out) if count_out is not None else None, _ptr(ws), C.c_size_t(nbytes), _stream(dev))
    return bins


def saturation_filter(order_sub: Tensor, reach_records: Tensor, dummy_index: int, tile_flags: Tensor, tile_bounds,
                      stats_out: Optional[Tensor] = None) -> Tensor:
    """``gsr_saturation_filter``: -> a copy of ``order_sub`` in which every Gaussian whose tile box holds no flagged
    tile is replaced by ``dummy_index`` (the row of a culled record, see ``count_reach(extra_rows=1)``).
    ``tile_flags`` int32[T]; ``stats_out`` int32[2] on the device: flagged tiles, Gaussians kept."""
    _check(order_sub, "order", _i32)
    _check(tile_flags, "tile_flags", _i32)
    dev = tile_flags.device
    with _on(dev):
        out = torch.empty_like(order_sub)
        nbytes = int(_lib().gsr_saturation_filter_workspace_bytes(C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1])))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _call("gsr_saturation_filter", C.c_int(order_sub.numel()), _ptr(order_sub), _ptr(reach_records),
              C.c_int(int(dummy_index)), _ptr(tile_flags), C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), _ptr(out),
              _ptr(ws), C.c_size_t(nbytes), _ptr(stats_out) if stats_out is not None else None, _stream(dev))
    return out


def rasterize_forward_round(rnd: int, tile_bounds, img_size, gaussian_ids_sorted, tile_bins, idx_base: int, xys, conics,
                            colors, extra, opacities, background, extra_background: float, out_img, out_extra, final_Ts,
                            final_idx, tile_flags, out_alpha=None, zero=None) -> None:
    """``gsr_rasterize_forward_round``: round 1 composites the prefix lists into RAW state (``final_Ts`` signed,
    ``out_img`` / ``out_extra`` without background) and flags the tiles with a live pixel; round 2 resumes over the
    second lists (``tile_bins`` relative to ``idx_base``) and finalises.  All outputs are caller-owned tensors;
    ``tile_flags`` (int32[T], zeroed before round 1) and ``out_alpha`` go to BOTH rounds."""
    W, H = int(img_size[0]), int(img_size[1])
    dev = xys.device
    nt = tile_bounds[0] * tile_bounds[1]
    with _on(dev):
        _call("gsr_rasterize_forward_round", C.c_int(int(rnd)), C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]),
              C.c_uint(W), C.c_uint(H), _ptr(gaussian_ids_sorted), _ptr(tile_bins), C.c_int(int(idx_base)), _ptr(xys),
              _ptr(conics), _ptr(colors), _ptr(extra) if extra is not None else None, _ptr(opacities),
              _ptr(background), C.c_float(extra_background), _ptr(out_img),
              _ptr(out_extra) if out_extra is not None else None, _ptr(final_Ts), _ptr(final_idx),
              _ptr(tile_flags), C.c_int(deep_tile_threshold(tile_bins.shape[0] * 400, nt)),
              _ptr(out_alpha) if out_alpha is not None else None, _ptr(zero) if zero is not None else None,
              C.c_size_t(zero.numel() * 4 if zero is not None else 0), _stream(dev))


def rasterize_backward_two(img_height, img_width, gaussian_ids_sorted, tile_bins, tile_bins2, idx_base2: int, xys,
                           conics, colors, extra, opacities, background, extra_background: float, final_Ts, final_idx,
                           v_output, v_output_extra, v_output_alpha, accumulators: Optional[Tensor] = None):
    """``gsr_rasterize_backward_two``: the backward over two-round lists -> (v_xy, v_conic, v_colors, v_opacity)
    or, with ``extra``, (v_xy, v_conic, v_colors, v_extra, v_opacity)."""
    v_output = _check(v_output.contiguous(), "v_output", _f32)
    if v_output_alpha is not None:
        v_output_alpha = _check(v_output_alpha.contiguous(), "v_output_alpha", _f32)
    if extra is not None:
        v_output_extra = _check(v_output_extra.contiguous(), "v_output_extra", _f32)
    n = xys.size(0)
    k = 10 if extra is not None else 9
    dev = xys.device
    with _on(dev):
        zeroed = accumulators is not None
        if zeroed and (accumulators.numel() != n * k or accumulators.dtype != _f32 or not accumulators.is_contiguous()):
            raise RuntimeError("rasterize_backward_two: accumulators must be backward_accumulators(n, 3 or 4, device)")
        flat = accumulators if zeroed else torch.empty((n * k,), dtype=_f32, device=dev)
        v_xy, v_conic = flat[: 2 * n].view(n, 2), flat[2 * n: 5 * n].view(n, 3)
        v_colors, v_opacity = flat[5 * n: 8 * n].view(n, 3), flat[8 * n: 9 * n].view(n, 1)
        v_extra = flat[9 * n:] if extra is not None else None
        nt = ((img_width + 15) // 16) * ((img_height + 15) // 16)
        _call("gsr_rasterize_backward_two", C.c_uint(img_height), C.c_uint(img_width), C.c_int(n),
              _ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(tile_bins2), C.c_int(int(idx_base2)), _ptr(xys),
              _ptr(conics), _ptr(colors), _ptr(extra) if extra is not None else None, _ptr(opacities),
              _ptr(background), C.c_float(extra_background), _ptr(final_Ts), _ptr(final_idx), _ptr(v_output),
              _ptr(v_output_extra) if extra is not None else None,
              _ptr(v_output_alpha) if v_output_alpha is not None else None, _ptr(v_xy), _ptr(v_conic), _ptr(v_colors),
              _ptr(v_extra) if v_extra is not None else None, _ptr(v_opacity),
              C.c_int(deep_tile_threshold(tile_bins.shape[0] * 400, nt, backward=True)), C.c_int(1 if zeroed else 0),
              _stream(dev))
    if extra is not None:
        return v_xy, v_conic, v_colors, v_extra, v_opacity
    return v_xy, v_conic, v_colors, v_opacity


def _raster_inputs(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background):
    _check(gaussian_ids_sorted, "gaussian_ids_sorted", _i32)
    _check(tile_bins, "tile_bins", _i32)
    for t, nm in ((xys, "xys"), (conics, "conics"), (colors, "colors"), (opacities, "opacities"),
                  (background, "background")):
        _check(t, nm, _f32)
    if xys.dim() != 2 or xys.size(1) != 2:
        raise RuntimeError("xys must have dimensions (num_points, 2)")
    if colors.dim() != 2:
        raise RuntimeError("colors must have 2 dimensions")
    if background.numel() != colors.size(1):
        raise RuntimeError("background must have one value per channel")


def depth_segments(list_entries: int, num_tiles: int):
    """-> (segments, minimum entries) for ``gsr_rasterize_forward_seg`` / ``gsr_rasterize_backward_seg``: into how many
    runs the lists of the tiles that are split over four waves are cut, each run walked by its own waves (DESIGN.md
    4.16).  Only tile grids that cannot fill the chip gain: 16 runs (`depth_segments` in _tuning.py; 1 = off) on grids of up to
    `depth_segments_grid` = 1 100 tiles (the grids on which forward and backward split every tile), for lists of more
    than `depth_segments_min` = 512 entries.  Measured (tools/exp/seg_ab.py, profiles/r04_depth_segments.txt):
    300 k Gaussians of the trainer's object scene at 480 x 270, compositing backward 436 -> 263 us with 8 runs
    (pre-pass 75 + walk 188), forward 338 -> 295 us with 4; config 3 (whose first 2 000 iterations run on this grid)
    818 / 825 -> 871 (8 runs) -> 885 iterations/s (16).  On larger grids the kernels are bound by their total work,
    which the pre-passes raise: 960 x 540 unchanged, the long-tail 1080p scene 0.62 -> 0.67 ms (backward), the default
    0.427 -> 0.448 ms (the empty workgroups of the segment grid) -- off there."""
    segs, grid, least = _segment_knobs()[:3]
    _, _, small_grid, _, small_grid_bwd = _deep_knobs()[:5]
    # (only where EVERY tile above the small-grid floor is split, forward and backward alike: which tiles are cut, and
    #  where, is then a function of the tile's list alone and every route to the kernels rounds the same way)
    if segs < 2 or num_tiles <= 0 or num_tiles > min(grid, small_grid, small_grid_bwd):
        return 1, 0
    return segs, least


def _forward_segments(list_entries: int, num_tiles: int, H: int, W: int, dev):
    """-> (segments, minimum entries, workspace or None) of ``gsr_rasterize_forward_seg`` for this tile grid."""
    segs, seg_min = depth_segments(list_entries, num_tiles)
    cap = _segment_knobs()[3]
    if segs > 1 and cap > 0:
        # `depth_segments_fwd` (_tuning.py): until round 6 the forward walked every list twice (a transmittance pre-pass
        # from T = 1 to each run's own saturation, then the runs) and a saturating scene paid for many runs (300 k opaque
        # Gaussians at 480 x 270: 338 us single, 295 with 4 runs, 432 with 16) while a translucent one gained all the way
        # (the model config 3 ends with: 0.54 / 0.39 / 0.31 ms with 4 / 8 / 16).  The forward now walks every run once
        # (csrc/raster_fwd.hip) and re-walks only the runs in which pixels cross the stop rule's threshold.
        segs = min(segs, cap)
    if segs < 2:
        return 0, 0, None
    nbytes = int(_lib().gsr_rasterize_forward_seg_workspace_bytes(C.c_uint(H), C.c_uint(W), C.c_int(segs)))
    return segs, seg_min, torch.empty((nbytes,), dtype=torch.uint8, device=dev)


_segment_cache = {}


def _segment_knobs():
    if not _segment_cache:
        # (the C entries take at most 16 runs -- GSR_REQUIRE(segments <= 16) -- and the backward calls them from inside
        #  autograd: a larger value in the table's overrides is clamped here instead of raising there)
        _segment_cache["v"] = (min(16, max(1, int(_tuning.get("depth_segments")))), int(_tuning.get("depth_segments_grid")),
                               int(_tuning.get("depth_segments_min")), min(16, max(0, int(_tuning.get("depth_segments_fwd")))))
    return _segment_cache["v"]


def deep_tile_threshold(list_entries: int, num_tiles: int, backward: bool = False) -> int:
    """List length above which a 16x16 tile is composited by four waves (one per 8x8
    sub-tile) instead of one (include/gsraster.h, ``deep_tile_threshold``): `deep_factor`
    (default 1.2; 0 = off) times the mean list length, not below `deep_min` (256; 1024 until round 5).
    `list_entries` is normally the capacity of device-sized lists (~1.25x the real count), so
    the default splits tiles above ~1.5x the mean.  Measured on the long-tail bench scene
    (10 % of the tiles ~10x deeper): forward 357 -> 320 us, backward 664 -> 625 us; factors
    0.8-1.5 within 3 % of each other, 0.3 (nearly every tile split) 1.7x slower; no effect
    on the uniform scene (nothing above the threshold; the idle workgroups cost < 1 %).
    Backward: `deep_factor_bwd` (2.0, quoted on a 1080p grid) scaled by tiles / 8 160 on grids above
    `small_grid_bwd` -- i.e. a tile is split when its list exceeds total entries / 4 096, the share of one of
    the backward's resident wave slots (`deep_factor_bwd_scaled`=0: the fixed factor)."""
    factor, floor, small_grid, small_floor, small_grid_bwd = _deep_knobs()[:5]
    if backward and _deep_knobs()[5] > 0:
        # `deep_factor_bwd` (2.0; 0 = the forward's factor): with the longest jobs first, the backward gains from
        # splitting only its longest tiles (four sub-tile waves run four butterflies): trained model 0.405 (1.2) ->
        # 0.348 ms (2.0) -> 0.40 (3.0+); long-tail scene 0.588 -> 0.559 -> 0.504 (6.0)
        factor = _deep_knobs()[5]
        if _deep_knobs()[6] and num_tiles > small_grid_bwd:
            # ... at 1080p.  The best factor follows the grid (trained model rendered at five sizes, job order on,
            # profiles/r05_midgrid_factors.txt): 2 040 tiles 0.5-0.7, 3 600 0.7-1.0, 4 590 1.4, 8 160 2.0, 14 400 3.0+
            # -- one per 4 096 tiles, the backward's resident waves (4 per SIMD): a tile is split when its list is
            # longer than the share of the launch's entries one wave slot would get.  960 x 540, the middle stage of
            # the reference's coarse-to-fine schedule: backward 0.637 -> 0.352 ms, config 3 879 -> 930+ iterations/s.
            factor *= num_tiles / _BWD_FACTOR_GRID
    if factor <= 0 or num_tiles <= 0:
        return 0
    if num_tiles <= (small_grid_bwd if backward else small_grid):
        # A small tile grid cannot fill the chip with one wave per tile (480 x 270 -- the first 2 000 iterations of
        # the reference's coarse-to-fine schedule -- is 510 tiles for 1 024 SIMDs) and every tile's list is long: the
        # walk is a serial chain of ~100 instructions per splat on a SIMD that has nothing else to issue.  Every
        # tile with more than a chunk or two of entries is split over four waves (one per 8x8 sub-tile).  Measured
        # (profiles/r04_small_grids.txt): 300 k Gaussians at 480 x 270, forward 0.32 -> 0.17 ms, backward 0.45 ->
        # 0.26 ms; 450 k at 960 x 540 (2 040 tiles), forward 0.19 -> 0.15 ms but backward 0.28 -> 0.35 ms (four
        # waves per tile also issue four times the atomics): the backward splits every tile only on grids of up to
        # `small_grid_bwd` tiles.  1080p with every tile split: 1.57 ms instead of 1.00.
        return small_floor
    return max(floor, int(factor * list_entries / num_tiles))


GSR_DEEP_ORDERED, GSR_DEEP_PREBUILT, GSR_DEEP_SECOND = 1 << 30, 1 << 29, 1 << 28  # include/gsraster.h
_DEEP_TAIL_SHIFT, _DEEP_THRESHOLD_MASK = 22, 0x3FFFFF


def tile_jobs_ints(tile_bounds) -> int:
    return int(_lib().gsr_tile_jobs_ints(C.c_int(int(tile_bounds[0])), C.c_int(int(tile_bounds[1]))))


_jobs_ints_cache = {}


def alloc_tile_bins(tile_bounds, dev) -> Tensor:
    """tile_bins [tiles, 2] int32 with room behind it for the launch's job order (include/gsraster.h,
    GSR_DEEP_ORDERED): one allocation; the returned tensor is the leading [tiles, 2] view."""
    key = (int(tile_bounds[0]), int(tile_bounds[1]))
    ints = _jobs_ints_cache.get(key)
    if ints is None:
        ints = _jobs_ints_cache[key] = tile_jobs_ints(key)
    nt = key[0] * key[1]
    bins = torch.empty((2 * nt + ints,), dtype=_i32, device=dev)[:2 * nt].view(nt, 2)
    bins._gsr_job_tail = ints > 0
    return bins


def deep_arg(tile_bins: Optional[Tensor], list_entries: int, num_tiles: int, backward: bool = False, tile_bounds=None) -> int:
    """The `deep_tile_threshold` argument of a compositing entry: the threshold (`deep_tile_threshold`), with
    GSR_DEEP_ORDERED set when `tile_bins` came from `alloc_tile_bins` (so the job order fits behind it), the grid
    is not a small one and `deep_order` is not 0 -- the entry then runs the launch's jobs longest first (DESIGN.md
    section 4.18)."""
    deep = deep_tile_threshold(list_entries, num_tiles, backward) if backward else deep_tile_threshold(list_entries, num_tiles)
    if deep <= 0 or tile_bins is None or tile_bounds is None or not _order_knob():
        return deep
    # small grids (every tile split, depth segments): the launch is a few hundred short jobs per run -- nothing to
    # order, and the step is bound by the host, where one more launch costs what it costs
    if num_tiles <= (_order_cache["grid"] if _order_cache["grid"] >= 0 else _deep_knobs()[4]):
        return deep
    key = (int(tile_bounds[0]), int(tile_bounds[1]))
    ints = _jobs_ints_cache.get(key)
    if ints is None:
        ints = _jobs_ints_cache[key] = tile_jobs_ints(key)
    if ints == 0:
        return deep  # (a grid beyond the order kernel's tables -- above 3840 x 2160: the static order)
    ok = getattr(tile_bins, "_gsr_job_tail", None)  # (decided once per tensor object)
    if ok is None:
        # A tensor that does not carry the attribute (unpacked from autograd's saved tensors: a new Python object over
        # the same storage; or a caller's own bins) owns a job tail only if `alloc_tile_bins` made its storage: the
        # whole storage is exactly  2 tiles + ints  int32 and the tensor starts at its beginning.  Room behind a slice
        # of some larger live buffer is NOT a tail -- the order kernel would write over whatever follows.
        ok = (tile_bins.dtype == _i32 and tile_bins.is_contiguous() and tile_bins.storage_offset() == 0
              and tile_bins.numel() == 2 * num_tiles
              and tile_bins.untyped_storage().nbytes() == 4 * (2 * num_tiles + ints))
        try:
            tile_bins._gsr_job_tail = ok
        except AttributeError:
            pass
    if not ok:
        return deep
    # (the forward's order lives in the first of the two arrays behind tile_bins, the backward's in the second)
    return (min(deep, _DEEP_THRESHOLD_MASK) | GSR_DEEP_ORDERED | (GSR_DEEP_SECOND if backward else 0)
            | (_order_cache["tail_bwd" if backward else "tail"] << _DEEP_TAIL_SHIFT))


def forward_orders(tile_bins: Optional[Tensor], list_entries: int, num_tiles: int, tile_bounds, dev) -> int:
    """`deep_arg` for a compositing FORWARD, with both job orders -- this launch's and the coming backward's -- written
    by ONE small launch (`gsr_tile_jobs_build`) in front of it: the backward then finds its order ready
    (`backward_order`) and launches nothing."""
    fwd = deep_arg(tile_bins, list_entries, num_tiles, tile_bounds=tile_bounds)
    if not (fwd & GSR_DEEP_ORDERED):
        return fwd
    bwd = deep_arg(tile_bins, list_entries, num_tiles, backward=True, tile_bounds=tile_bounds)
    second = bwd if (bwd & GSR_DEEP_ORDERED) else 0
    if second and getattr(tile_bins, "_gsr_jobs_fwd", 0) == fwd and getattr(tile_bins, "_gsr_jobs_bwd", 0) == second:
        return fwd | GSR_DEEP_PREBUILT  # (the call that built these lists has written both orders behind them)
    _call("gsr_tile_jobs_build", C.c_int(int(tile_bounds[0])), C.c_int(int(tile_bounds[1])), _ptr(tile_bins), C.c_int(fwd),
          C.c_int(second), _stream(dev))
    try:
        tile_bins._gsr_jobs_bwd, tile_bins._gsr_jobs_fwd = second, (fwd if second else 0)
    except AttributeError:
        pass
    return fwd | GSR_DEEP_PREBUILT


def backward_order(tile_bins: Optional[Tensor], list_entries: int, num_tiles: int, tile_bounds) -> int:
    """`deep_arg` for a compositing BACKWARD: with GSR_DEEP_PREBUILT when the forward of these very lists has already
    written this order (`forward_orders` left the argument it built for on the tensor), else the entry builds it."""
    arg = deep_arg(tile_bins, list_entries, num_tiles, backward=True, tile_bounds=tile_bounds)
    if (arg & GSR_DEEP_ORDERED) and getattr(tile_bins, "_gsr_jobs_bwd", 0) == arg:
        arg |= GSR_DEEP_PREBUILT
    return arg


_order_cache = {}


def _order_knob() -> bool:
    if not _order_cache:
        _order_cache["v"] = bool(int(_tuning.get("deep_order")))
        # the share (in 1/64ths) of a launch's whole-tile jobs that run last as four sub-tile jobs each (csrc/raster_common.h)
        _order_cache["tail"] = min(63, max(0, int(_tuning.get("deep_tail"))))
        _order_cache["tail_bwd"] = min(63, max(0, int(_tuning.get("deep_tail_bwd"))))
        # grids of up to this many tiles run in the static order (-1: small_grid_bwd, the grids on which both
        # directions split every tile and cut the lists into depth segments)
        _order_cache["grid"] = int(_tuning.get("deep_order_grid"))
    return _order_cache["v"]


_deep_cache = {}
_BWD_FACTOR_GRID = 8160.0  # (the grid `deep_factor_bwd` is quoted on: 1920 x 1080)


def _deep_knobs():
    # (read once: table look-ups per compositing call are measurable on small scenes)
    if not _deep_cache:
        g = _tuning.get
        _deep_cache["v"] = (float(g("deep_factor")), int(g("deep_min")), int(g("small_grid")), int(g("small_grid_min")),
                            int(g("small_grid_bwd")), float(g("deep_factor_bwd")), bool(int(g("deep_factor_bwd_scaled"))))
    return _deep_cache["v"]


@_tuning.on_change
def _drop_tuning_caches():
    _segment_cache.clear()
    _order_cache.clear()
    _deep_cache.clear()


class _RasterDesc(C.Structure):  # gsr_raster_desc (include/gsraster.h)
    _fields_ = ([(k, C.c_int) for k in ("num_points", "img_height", "img_width", "capacity", "deep_tile_threshold")]
                + [("extra_background", C.c_float)]
                + [(k, C.c_void_p) for k in ("xys", "depths", "radii", "conics", "colors", "extra", "opac", "background",
                                             "order_ready", "reach_records", "counts", "order", "cum", "ids",
                                             "tile_bins", "count_out", "sort_ws")]
                + [("sort_ws_bytes", C.c_size_t), ("bin_ws", C.c_void_p), ("bin_ws_bytes", C.c_size_t)]
                + [(k, C.c_void_p) for k in ("out_img", "out_extra", "final_Ts", "final_idx", "out_alpha", "zero_ptr")]
                + [("zero_bytes", C.c_size_t), ("segments", C.c_int), ("segment_min_entries", C.c_int),
                   ("seg_ws", C.c_void_p), ("seg_ws_bytes", C.c_size_t), ("deep_tile_threshold_backward", C.c_int)])


_raster_plan_cache = {}


def _raster_plan(n: int, capacity: int, tb, have_order: bool):
    """Byte offsets of the scratch regions of one `rasterize_gaussians_forward` call inside ONE allocation
    (records | order | counts | cum | sort workspace | partition workspace, each 256-byte aligned)."""
    key = (n, capacity, tb[0], tb[1], have_order)
    plan = _raster_plan_cache.get(key)
    if plan is None:
        lib = _lib()
        lean = not lists_need_counts(n, capacity, tb, device_sized=True)
        al = lambda b: (int(b) + 255) & ~255
        rec_b = al(n * int(lib.gsr_reach_record_bytes()))
        sort_b = 0 if (have_order and lean) else int(lib.gsr_depth_order_workspace_bytes(C.c_int(n), C.c_int(1)))
        bin_b = int(lib.gsr_bin_sorted_workspace_bytes(C.c_int(n), C.c_int(capacity), C.c_int(tb[0]), C.c_int(tb[1])))
        off, o = {}, 0
        for name, size in (("records", rec_b), ("order", 0 if (have_order and lean) else al(4 * n)),
                           ("counts", 0 if lean else al(4 * n)), ("cum", 0 if lean else al(4 * n)),
                           ("sort_ws", al(sort_b)), ("bin_ws", al(bin_b))):
            off[name] = (o, size)
            o += size
        plan = (lean, off, max(o, 256), sort_b, bin_b)
        if len(_raster_plan_cache) > 64:
            _raster_plan_cache.clear()
        _raster_plan_cache[key] = plan
    return plan


def rasterize_gaussians_forward(xys, depths, radii, conics, colors, opacities, background, img_height: int,
                                img_width: int, capacity: int, count_out: Tensor, want_alpha: bool = False, zero=None,
                                order_ready: Optional[Tensor] = None, extra: Optional[Tensor] = None,
                                extra_background: float = 0.0, composite: bool = True, checked: bool = False):
    """``gsr_rasterize_gaussians_forward``: reach records + depth order + device-sized tile lists + compositing of
    16x16 tiles / 3 channels in ONE native call (what ``_RasterizeGaussians.forward`` runs on the device,
    rasterize.py:89-170 of the reference) -> (gaussian_ids_sorted i32[capacity], tile_bins i32[T,2], out_img,
    final_Ts, final_idx, alpha or None[, out_extra]).  ``count_out`` (int32[1], pinned or device) receives the number
    of list entries the view needs: above ``capacity`` the lists were cut and the caller builds them again.
    ``order_ready``: the depth order of these depths / radii if the caller already has it (the sort is skipped).
    ``composite=False``: the lists only -> (gaussian_ids_sorted, tile_bins); ``colors`` / ``background`` unused."""
    n = xys.size(0)
    dev = xys.device
    for t, dt, width in () if checked else ((xys, _f32, 2), (depths, _f32, 1), (radii, _i32, 1), (conics, _f32, 3), (opacities, _f32, 1)) + (
            ((colors, _f32, 3), (background, _f32, None)) if composite else ()):
        if t.dtype != dt or t.device != dev or not t.is_contiguous() or (width is not None and t.numel() != n * width):
            raise RuntimeError("rasterize_gaussians_forward: float32 / int32 contiguous tensors of one device with N rows "
                               "expected (xys [N,2], depths [N], radii [N], conics [N,3], colors [N,3], opacities [N,1])")
    if (composite and background.numel() != 3) or n < 1 or capacity < 1:
        raise RuntimeError("rasterize_gaussians_forward: background [3], N >= 1 and capacity >= 1 expected")
    tb = ((img_width + 15) // 16, (img_height + 15) // 16, 1)
    if order_ready is not None and lists_need_counts(n, capacity, tb, device_sized=True):
        order_ready = None  # lists with counts sort the counts along: the ready-made order is of no use
    lean, off, total, sort_b, bin_b = _raster_plan(n, capacity, tb, order_ready is not None)
    H, W = int(img_height), int(img_width)
    with _on(dev):
        ws = torch.empty((total,), dtype=torch.uint8, device=dev)
        ids = torch.empty((capacity,), dtype=_i32, device=dev)
        bins = alloc_tile_bins(tb, dev)
        img = Ts = idx = alpha = out_extra = None
        if composite:
            img = torch.empty((H, W, 3), dtype=_f32, device=dev)
            # final_Ts | final_idx | alpha [| extra]: one allocation, three (four) [H,W] planes
            planes = torch.empty((2 + int(want_alpha) + int(extra is not None), H, W), dtype=_f32, device=dev)
            Ts, idx = planes[0], planes[1].view(_i32)
            alpha = planes[2] if want_alpha else None
            out_extra = planes[2 + int(want_alpha)] if extra is not None else None
        base = ws.data_ptr()
        at = lambda name: (base + off[name][0]) if off[name][1] else None
        p = lambda t: None if t is None else t.data_ptr()
        zero_bytes = 0
        if zero is not None:
            zero_bytes = zero.numel() * 4
            if zero_bytes == 0:
                zero = None
        segs, seg_min, seg_ws = _forward_segments(capacity, tb[0] * tb[1], H, W, dev) if composite else (0, 0, None)
        desc = _RasterDesc(n, H, W, int(capacity), deep_arg(bins, capacity, tb[0] * tb[1], tile_bounds=tb), float(extra_background),
                           xys.data_ptr(), depths.data_ptr(), radii.data_ptr(), conics.data_ptr(),
                           p(colors) if composite else None, p(extra), opacities.data_ptr(),
                           p(background) if composite else None, p(order_ready), at("records"),
                           at("counts"), at("order"), at("cum"), ids.data_ptr(), bins.data_ptr(), count_out.data_ptr(),
                           at("sort_ws"), sort_b, at("bin_ws"), bin_b, p(img), p(out_extra), p(Ts),
                           p(idx), p(alpha), p(zero), zero_bytes, segs, seg_min, p(seg_ws),
                           seg_ws.numel() if seg_ws is not None else 0, 0)
        if desc.deep_tile_threshold & GSR_DEEP_ORDERED:
            # both job orders -- this forward's (or, for lists built ahead, the later compositing call's) and the
            # coming backward's -- ride in one launch behind the lists (include/gsraster.h)
            bwd = deep_arg(bins, capacity, tb[0] * tb[1], backward=True, tile_bounds=tb)
            if bwd & GSR_DEEP_ORDERED:
                desc.deep_tile_threshold_backward = bwd
                bins._gsr_jobs_bwd = bwd
                bins._gsr_jobs_fwd = int(desc.deep_tile_threshold)
        _call("gsr_rasterize_gaussians_forward", C.byref(desc), _stream(dev))
    if not composite:
        return ids, bins
    if extra is not None:
        return ids, bins, img, Ts, idx, alpha, out_extra
    return ids, bins, img, Ts, idx, alpha


def composite_prepared(tile_bounds, img_width: int, img_height: int, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                       opacities, background, out_img, planes, want_alpha: bool, zero=None):
    """``gsr_rasterize_forward_ex`` into caller-owned outputs, for callers that have validated their tensors and
    allocated ``out_img`` [H,W,3] and ``planes`` [3,H,W] (final_Ts | final_idx as int32 | alpha) ahead of time: the
    shortest host path to the compositing launch (rasterize.py, "lists built ahead of time")
    -> (out_img, final_Ts, final_idx, alpha or None)."""
    dev = xys.device
    Ts, idx = planes[0], planes[1].view(_i32)
    alpha = planes[2] if want_alpha else None
    zero_bytes = zero.numel() * 4 if zero is not None else 0
    with _on(dev):
        tiles = tile_bounds[0] * tile_bounds[1]
        segs, seg_min, seg_ws = _forward_segments(gaussian_ids_sorted.numel(), tiles, int(img_height), int(img_width), dev)
        _call("gsr_rasterize_forward_seg", C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]),
              C.c_uint(int(img_width)), C.c_uint(int(img_height)), _ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(xys),
              _ptr(conics), _ptr(colors), None, _ptr(opacities), _ptr(background), C.c_float(0.0), _ptr(out_img), None,
              _ptr(Ts), _ptr(idx), C.c_int(forward_orders(tile_bins, gaussian_ids_sorted.numel(), tiles, tile_bounds, dev)),
              _ptr(alpha) if alpha is not None else None, _ptr(zero) if zero_bytes else None, C.c_size_t(zero_bytes),
              C.c_int(segs), C.c_int(seg_min), _ptr(seg_ws) if seg_ws is not None else None,
              C.c_size_t(seg_ws.numel() if seg_ws is not None else 0), _stream(dev))
    return out_img, Ts, idx, alpha


def rasterize_forward_ex(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics,
                         colors, opacities, background, want_alpha=False, zero=None):
    """``gsr_rasterize_forward_ex`` (16x16 tiles, 3 channels): -> (out_img, final_Ts, final_idx, alpha or None);
    ``alpha = 1 - final_Ts`` written by the kernel; ``zero``: a float32 tensor the launch clears (the
    accumulators of the coming :func:`rasterize_backward` -- see ``accumulators``)."""
    return _rasterize_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys,
                              conics, colors, opacities, background, nd=False, want_alpha=want_alpha, zero=zero,
                              ex=True)


def _rasterize_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics,
                       colors, opacities, background, nd: bool, want_alpha: bool = False, zero=None, ex: bool = False):
    _raster_inputs(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background)
    channels = colors.size(1)
    W, H = int(img_size[0]), int(img_size[1])
    dev = xys.device
    with _on(dev):
        out_img = torch.empty((H, W, channels), dtype=_f32, device=dev)
        final_Ts = torch.empty((H, W), dtype=_f32, device=dev)
        final_idx = torch.empty((H, W), dtype=_i32, device=dev)
        head = (C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), C.c_uint(block[0]), C.c_uint(W),
                C.c_uint(H))
        tail = (_ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(xys), _ptr(conics), _ptr(colors),
                _ptr(opacities), _ptr(background), _ptr(out_img), _ptr(final_Ts), _ptr(final_idx))
        if nd:
            _call("gsr_rasterize_forward_nd", *head, C.c_uint(channels), *tail, _stream(dev))
        else:
            if channels != 3:
                raise RuntimeError("rasterize_forward expects 3 channels; use nd_rasterize_forward")
            deep = forward_orders(tile_bins, gaussian_ids_sorted.numel(), tile_bounds[0] * tile_bounds[1], tile_bounds, dev) \
                if block[0] == 16 else deep_tile_threshold(gaussian_ids_sorted.numel(), tile_bounds[0] * tile_bounds[1])
            segs, seg_min, seg_ws = _forward_segments(gaussian_ids_sorted.numel(), tile_bounds[0] * tile_bounds[1], H, W,
                                                      dev) if block[0] == 16 else (0, 0, None)
            if ex or segs > 1:
                if zero is not None:
                    _check(zero, "zero", _f32)
                    if zero.numel() == 0:
                        zero = None
                alpha = torch.empty((H, W), dtype=_f32, device=dev) if want_alpha else None
                tail_ex = (C.c_int(deep), _ptr(alpha) if alpha is not None else None,
                           _ptr(zero) if zero is not None else None,
                           C.c_size_t(zero.numel() * 4 if zero is not None else 0))
                if segs > 1:
                    _call("gsr_rasterize_forward_seg", head[0], head[1], head[3], head[4], *tail[:5], None, *tail[5:7],
                          C.c_float(0.0), tail[7], None, *tail[8:], *tail_ex, C.c_int(segs), C.c_int(seg_min),
                          _ptr(seg_ws), C.c_size_t(seg_ws.numel()), _stream(dev))
                else:
                    _call("gsr_rasterize_forward_ex", *head, *tail, *tail_ex, _stream(dev))
                return (out_img, final_Ts, final_idx, alpha) if ex else (out_img, final_Ts, final_idx)
            _call("gsr_rasterize_forward", *head, *tail, C.c_int(deep), _stream(dev))
    return out_img, final_Ts, final_idx


def rasterize_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics,
                      colors, opacities, background):
    """-> (out_img [H,W,3], final_Ts [H,W], final_idx i32[H,W]);
    replaces ``rasterize_forward_tensor`` (bindings.cu:269-328)."""
    return _rasterize_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys,
                              conics, colors, opacities, background, nd=False)


def rasterize_forward_scan(tile_bounds, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities,
                           background):
    """`rasterize_forward` through the scan mapping (``gsr_rasterize_forward_scan``; measurement variant,
    16x16 tiles): lanes over splats, wave prefix product for the transmittance."""
    _raster_inputs(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background)
    W, H = int(img_size[0]), int(img_size[1])
    dev = xys.device
    with _on(dev):
        img = torch.empty((H, W, 3), dtype=_f32, device=dev)
        Ts = torch.empty((H, W), dtype=_f32, device=dev)
        idx = torch.empty((H, W), dtype=_i32, device=dev)
        _call("gsr_rasterize_forward_scan", C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), C.c_uint(W), C.c_uint(H),
              _ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(xys), _ptr(conics), _ptr(colors), _ptr(opacities),
              _ptr(background), _ptr(img), _ptr(Ts), _ptr(idx), _stream(dev))
    return img, Ts, idx


def nd_rasterize_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics,
                         colors, opacities, background):
    """Generic channel count; replaces ``nd_rasterize_forward_tensor``
    (bindings.cu:330-399).  Accumulates in fp32 (the reference uses fp16)."""
    return _rasterize_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys,
                              conics, colors, opacities, background, nd=True)


def rasterize_forward_rgbd(tile_bounds, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors, extra,
                           opacities, background, extra_background: float, want_alpha: bool = False, zero=None):
    """RGB + one extra channel in one pass (``gsr_rasterize_forward_rgbd``, 16x16 tiles):
    -> (out_img [H,W,3], out_extra [H,W], final_Ts [H,W], final_idx i32[H,W]) [+ alpha = 1 - final_Ts
    when ``want_alpha``].  ``zero``: a float32 tensor the launch clears (``backward_accumulators(n, 4, dev)``
    for the coming :func:`rasterize_backward_rgbd`)."""
    _raster_inputs(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background)
    _check(extra, "extra", _f32)
    if colors.size(1) != 3 or extra.numel() != xys.size(0):
        raise RuntimeError("rasterize_forward_rgbd expects colors [N,3] and extra [N]")
    W, H = int(img_size[0]), int(img_size[1])
    dev = xys.device
    with _on(dev):
        img = torch.empty((H, W, 3), dtype=_f32, device=dev)
        ext = torch.empty((H, W), dtype=_f32, device=dev)
        Ts = torch.empty((H, W), dtype=_f32, device=dev)
        idx = torch.empty((H, W), dtype=_i32, device=dev)
        alpha = torch.empty((H, W), dtype=_f32, device=dev) if want_alpha else None
        if zero is not None:
            _check(zero, "zero", _f32)
        segs, seg_min, seg_ws = _forward_segments(gaussian_ids_sorted.numel(), tile_bounds[0] * tile_bounds[1], H, W, dev)
        _call("gsr_rasterize_forward_seg", C.c_int(tile_bounds[0]), C.c_int(tile_bounds[1]), C.c_uint(W),
              C.c_uint(H), _ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(xys), _ptr(conics), _ptr(colors),
              _ptr(extra), _ptr(opacities), _ptr(background), C.c_float(extra_background), _ptr(img), _ptr(ext),
              _ptr(Ts), _ptr(idx),
              C.c_int(forward_orders(tile_bins, gaussian_ids_sorted.numel(), tile_bounds[0] * tile_bounds[1], tile_bounds, dev)),
              _ptr(alpha) if alpha is not None else None, _ptr(zero) if zero is not None else None,
              C.c_size_t(zero.numel() * 4 if zero is not None else 0), C.c_int(segs), C.c_int(seg_min),
              _ptr(seg_ws) if seg_ws is not None else None, C.c_size_t(seg_ws.numel() if seg_ws is not None else 0),
              _stream(dev))
    if want_alpha:
        return img, ext, Ts, idx, alpha
    return img, ext, Ts, idx


def rasterize_backward_rgbd(img_height, img_width, gaussian_ids_sorted, tile_bins, xys, conics, colors, extra,
                            opacities, background, extra_background, final_Ts, final_idx, v_output,
                            v_output_extra, v_output_alpha, accumulators: Optional[Tensor] = None):
    """-> (v_xy, v_conic, v_colors, v_extra [N], v_opacity [N,1]); ``gsr_rasterize_backward_rgbd``.
    ``accumulators``: ``backward_accumulators(n, 4, dev)`` already cleared by the forward launch."""
    _raster_inputs(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background)
    _check(extra, "extra", _f32)
    v_output = _check(v_output.contiguous(), "v_output", _f32)
    v_output_extra = _check(v_output_extra.contiguous(), "v_output_extra", _f32)
    if v_output_alpha is not None:
        v_output_alpha = _check(v_output_alpha.contiguous(), "v_output_alpha", _f32)
    n = xys.size(0)
    dev = xys.device
    with _on(dev):
        if accumulators is not None and (accumulators.numel() != n * 10 or accumulators.dtype != _f32 or
                                         not accumulators.is_contiguous()):
            raise RuntimeError("rasterize_backward_rgbd: accumulators must be backward_accumulators(n, 4, device)")
        flat = accumulators if accumulators is not None else torch.empty((n * 10,), dtype=_f32, device=dev)
        v_xy, v_conic = flat[: 2 * n].view(n, 2), flat[2 * n: 5 * n].view(n, 3)
        v_colors, v_opacity = flat[5 * n: 8 * n].view(n, 3), flat[8 * n: 9 * n].view(n, 1)
        v_extra = flat[9 * n:]
        tiles = ((img_width + 15) // 16) * ((img_height + 15) // 16)
        segs, seg_min = depth_segments(gaussian_ids_sorted.numel(), tiles)
        ws = torch.empty(((segs - 1) * int(img_height) * int(img_width), 2), dtype=_f32, device=dev) if segs > 1 else None
        _call("gsr_rasterize_backward_seg", C.c_uint(img_height), C.c_uint(img_width), C.c_int(n),
              _ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(xys), _ptr(conics), _ptr(colors), _ptr(extra),
              _ptr(opacities), _ptr(background), C.c_float(extra_background), _ptr(final_Ts), _ptr(final_idx),
              _ptr(v_output), _ptr(v_output_extra),
              _ptr(v_output_alpha) if v_output_alpha is not None else None, _ptr(v_xy), _ptr(v_conic),
              _ptr(v_colors), _ptr(v_extra), _ptr(v_opacity),
              C.c_int(backward_order(tile_bins, gaussian_ids_sorted.numel(), tiles,
                                     ((int(img_width) + 15) // 16, (int(img_height) + 15) // 16))),
              C.c_int(1 if accumulators is not None else 0), C.c_int(segs if ws is not None else 0), C.c_int(seg_min),
              _ptr(ws) if ws is not None else None, C.c_size_t(ws.numel() * 4 if ws is not None else 0), _stream(dev))
    return v_xy, v_conic, v_colors, v_extra, v_opacity


def rasterize_backward_det(img_height, img_width, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities,
                           background, final_Ts, final_idx, v_output, v_output_alpha, order, cum_sorted, slot_of_entry,
                           extra=None, extra_background: float = 0.0, v_output_extra=None):
    """``gsr_rasterize_backward_det``: the compositing backward with a fixed summation order
    (bit-identical gradients from run to run) -> (v_xy, v_conic, v_colors, v_opacity [N,1]) or,
    with ``extra`` / ``v_output_extra``, (v_xy, v_conic, v_colors, v_extra [N], v_opacity)."""
    _raster_inputs(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background)
    for t, nm in ((order, "order"), (cum_sorted, "cum_sorted"), (slot_of_entry, "slot_of_entry")):
        _check(t, nm, _i32)
    v_output = _check(v_output.contiguous(), "v_output", _f32)
    if v_output_alpha is not None:
        v_output_alpha = _check(v_output_alpha.contiguous(), "v_output_alpha", _f32)
    rgbd = extra is not None
    if rgbd:
        _check(extra, "extra", _f32)
        v_output_extra = _check(v_output_extra.contiguous(), "v_output_extra", _f32)
    n = xys.size(0)
    L = gaussian_ids_sorted.numel()
    if slot_of_entry.numel() != L or order.numel() != n or cum_sorted.numel() % max(n, 1):
        raise RuntimeError("rasterize_backward_det: order / cum_sorted / slot_of_entry do not match the lists")
    bands = cum_sorted.numel() // n if n else 1
    dev = xys.device
    _o = lambda t: None if t is None else _ptr(t)
    with _on(dev):
        v_xy = torch.empty((n, 2), dtype=_f32, device=dev)
        v_conic = torch.empty((n, 3), dtype=_f32, device=dev)
        v_colors = torch.empty((n, 3), dtype=_f32, device=dev)
        v_opacity = torch.empty((n, 1), dtype=_f32, device=dev)
        v_extra = torch.empty((n,), dtype=_f32, device=dev) if rgbd else None
        nbytes = int(_lib().gsr_rasterize_backward_det_workspace_bytes(C.c_int(L)))
        ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev)
        _call("gsr_rasterize_backward_det", C.c_uint(img_height), C.c_uint(img_width), C.c_int(n), C.c_int(L),
              _ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(xys), _ptr(conics), _ptr(colors), _o(extra),
              _ptr(opacities), _ptr(background), C.c_float(extra_background), _ptr(final_Ts), _ptr(final_idx),
              _ptr(v_output), _o(v_output_extra), _o(v_output_alpha), _ptr(order), _ptr(cum_sorted), C.c_int(bands),
              _ptr(slot_of_entry), _ptr(ws), C.c_size_t(nbytes), _ptr(v_xy), _ptr(v_conic), _ptr(v_colors),
              _o(v_extra), _ptr(v_opacity), _stream(dev))
    if rgbd:
        return v_xy, v_conic, v_colors, v_extra, v_opacity
    return v_xy, v_conic, v_colors, v_opacity


def backward_accumulators(n: int, channels: int, device) -> Tensor:
    """The flat float32 buffer :func:`rasterize_backward` carves ``v_xy | v_conic | v_colors | v_opacity``
    out of.  Cleared by ``rasterize_forward_ex(zero=...)`` and handed back as ``accumulators`` it spares
    the backward its own zero fill."""
    with torch.cuda.device(device):
        return torch.empty((n * (6 + channels),), dtype=_f32, device=device)


def _rasterize_backward(img_height, img_width, block_width, gaussian_ids_sorted, tile_bins, xys,
                        conics, colors, opacities, background, final_Ts, final_idx, v_output,
                        v_output_alpha, nd: bool, accumulators: Optional[Tensor] = None, trusted: bool = False):
    if not trusted:  # (trusted: everything but the two cotangents went through the forward's checks)
        _raster_inputs(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background)
        if not nd and colors.size(1) != 3:
            raise RuntimeError("colors must have 2 dimensions")  # message of bindings.cu:494-496
        _check(final_Ts, "final_Ts", _f32)
        _check(final_idx, "final_idx", _i32)
    v_output = _check(v_output.contiguous(), "v_output", _f32)
    if v_output_alpha is not None:  # None = zero cotangent for the alpha output
        v_output_alpha = _check(v_output_alpha.contiguous(), "v_output_alpha", _f32)
    n, channels = xys.size(0), colors.size(1)
    dev = xys.device
    with _on(dev):
        # four contiguous tensors carved out of one allocation: the library
        # zero-fills them with a single memset when they are back to back
        zeroed = accumulators is not None and not nd and block_width == 16
        if zeroed and (accumulators.numel() != n * (6 + channels) or accumulators.dtype != _f32 or
                       not accumulators.is_contiguous()):
            raise RuntimeError("rasterize_backward: accumulators must be backward_accumulators(n, channels, device)")
        flat = accumulators if zeroed else torch.empty((n * (6 + channels),), dtype=_f32, device=dev)
        v_xy = flat[: 2 * n].view(n, 2)
        v_conic = flat[2 * n: 5 * n].view(n, 3)
        v_colors = flat[5 * n: (5 + channels) * n].view(n, channels)
        v_opacity = flat[(5 + channels) * n:].view(n, 1)
        head = (C.c_uint(img_height), C.c_uint(img_width), C.c_uint(block_width))
        tail = (C.c_int(n), _ptr(gaussian_ids_sorted), _ptr(tile_bins), _ptr(xys), _ptr(conics),
                _ptr(colors), _ptr(opacities), _ptr(background), _ptr(final_Ts), _ptr(final_idx),
                _ptr(v_output), _ptr(v_output_alpha) if v_output_alpha is not None else None, _ptr(v_xy), _ptr(v_conic), _ptr(v_colors),
                _ptr(v_opacity))
        if nd:
            _call("gsr_rasterize_backward_nd", *head, C.c_uint(channels), *tail, _stream(dev))
        else:
            tiles = ((img_width + block_width - 1) // block_width) * ((img_height + block_width - 1) // block_width)
            segs, seg_min = depth_segments(gaussian_ids_sorted.numel(), tiles) if block_width == 16 else (1, 0)
            _tb16 = ((int(img_width) + 15) // 16, (int(img_height) + 15) // 16) if block_width == 16 else None
            if segs > 1:
                deep = backward_order(tile_bins, gaussian_ids_sorted.numel(), tiles,
                                      ((int(img_width) + 15) // 16, (int(img_height) + 15) // 16))
                ws = torch.empty(((segs - 1) * int(img_height) * int(img_width), 2), dtype=_f32, device=dev)
                _call("gsr_rasterize_backward_seg", C.c_uint(img_height), C.c_uint(img_width), *tail[:6], None,
                      *tail[6:8], C.c_float(0.0), *tail[8:11], None, *tail[11:15], None, tail[15], C.c_int(deep),
                      C.c_int(1 if zeroed else 0), C.c_int(segs), C.c_int(seg_min), _ptr(ws),
                      C.c_size_t(ws.numel() * 4), _stream(dev))
            elif zeroed:
                _call("gsr_rasterize_backward_ex", *head, *tail,
                      C.c_int(backward_order(tile_bins, gaussian_ids_sorted.numel(), tiles, _tb16) if _tb16 else
                              deep_tile_threshold(gaussian_ids_sorted.numel(), tiles, backward=True)),
                      C.c_int(1), _stream(dev))
            else:
                _call("gsr_rasterize_backward", *head, *tail,
                      C.c_int(backward_order(tile_bins, gaussian_ids_sorted.numel(), tiles, _tb16) if _tb16 else
                              deep_tile_threshold(gaussian_ids_sorted.numel(), tiles, backward=True)),
                      _stream(dev))
    return v_xy, v_conic, v_colors, v_opacity


def rasterize_backward(img_height, img_width, block_width, gaussian_ids_sorted, tile_bins, xys,
                       conics, colors, opacities, background, final_Ts, final_idx, v_output,
                       v_output_alpha, accumulators: Optional[Tensor] = None, trusted: bool = False):
    """-> (v_xy, v_conic, v_colors, v_opacity [N,1]);
    replaces ``rasterize_backward_tensor`` (bindings.cu:476-528).  ``accumulators``: a
    :func:`backward_accumulators` buffer already cleared by :func:`rasterize_forward_ex`."""
    return _rasterize_backward(img_height, img_width, block_width, gaussian_ids_sorted, tile_bins,
                               xys, conics, colors, opacities, background, final_Ts, final_idx,
                               v_output, v_output_alpha, nd=False, accumulators=accumulators, trusted=trusted)


def nd_rasterize_backward(img_height, img_width, block_width, gaussian_ids_sorted, tile_bins, xys,
                          conics, colors, opacities, background, final_Ts, final_idx, v_output,
                          v_output_alpha):
    """Generic channel count; replaces ``nd_rasterize_backward_tensor`` (bindings.cu:406-469)."""
    return _rasterize_backward(img_height, img_width, block_width, gaussian_ids_sorted, tile_bins,
                               xys, conics, colors, opacities, background, final_Ts, final_idx,
                               v_output, v_output_alpha, nd=True)


def compute_cov2d_bounds(num_pts: int, cov2d: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (conics [N,3], radii [N,1] f32); replaces ``compute_cov2d_bounds_tensor``
    (bindings.cu:39-56)."""
    _check(cov2d, "cov2d", _f32)
    n = int(num_pts)
    dev = cov2d.device
    with _on(dev):
        conics = torch.empty((n, cov2d.size(1)), dtype=_f32, device=dev)
        radii = torch.empty((n, 1), dtype=_f32, device=dev)
        _call("gsr_cov2d_bounds", C.c_int(n), _ptr(cov2d), _ptr(conics), _ptr(radii), _stream(dev))
    return conics, radii
