"""Loader of the native library behind ``rasterizer.cuda``.

Counterpart of the reference's ``rasterizer/cuda/_backend.py`` (prebuilt
extension first, nvcc JIT otherwise).  Here there is exactly one backend: the
HIP library ``libgsraster.so`` (C ABI in ``include/gsraster.h``) built in-tree
for gfx950 by ``csrc/Makefile``.  There is no CPU or eager-PyTorch fallback: if
the library is missing the first native call raises.
"""
import ctypes as C
import os
import subprocess

PATH = os.path.dirname(os.path.abspath(__file__))
# (GSR_LIBRARY: another build of the same library, for A/B measurements of compile-time variants -- tools/r05/)
LIB_PATH = os.environ.get("GSR_LIBRARY") or os.path.join(PATH, "libgsraster.so")
CSRC = os.path.normpath(os.path.join(PATH, "..", "..", "csrc"))

# every symbol include/gsraster.h declares
SYMBOLS = (
    "gsr_version",
    "gsr_last_error",
    "gsr_project_forward",
    "gsr_project_backward",
    "gsr_sh_forward",
    "gsr_sh_backward",
    "gsr_cumsum_workspace_bytes",
    "gsr_cumsum_tiles",
    "gsr_map_intersects",
    "gsr_sort_workspace_bytes",
    "gsr_sort_intersects",
    "gsr_tile_bin_edges",
    "gsr_reach_record_bytes",
    "gsr_tile_bands",
    "gsr_count_reach",
    "gsr_depth_order_workspace_bytes",
    "gsr_depth_order",
    "gsr_reach_records_depth_order",
    "gsr_bin_sorted_workspace_bytes",
    "gsr_bin_sorted_needs_counts",
    "gsr_bin_sorted",
    "gsr_bin_sorted_dev",
    "gsr_publish_int32",
    "gsr_rasterize_forward",
    "gsr_rasterize_backward",
    "gsr_rasterize_forward_ex",
    "gsr_rasterize_backward_ex",
    "gsr_rasterize_forward_seg",
    "gsr_rasterize_forward_seg_workspace_bytes",
    "gsr_rasterize_backward_seg",
    "gsr_rasterize_backward_seg_workspace_bytes",
    "gsr_rasterize_forward_nd",
    "gsr_rasterize_backward_nd",
    "gsr_cov2d_bounds",
    "gsr_l1_ssim_forward",
    "gsr_l1_ssim_backward",
    "gsr_l1_forward",
    "gsr_l1_backward",
    "gsr_depth_l1_forward",
    "gsr_depth_l1_backward",
    "gsr_sh_forward_split",
    "gsr_sh_backward_split",
    "gsr_sh_backward_views",
    "gsr_rasterize_forward_rgbd",
    "gsr_rasterize_forward_scan",
    "gsr_tile_lists_subrange_workspace_bytes",
    "gsr_tile_lists_subrange",
    "gsr_saturation_filter_workspace_bytes",
    "gsr_saturation_filter",
    "gsr_rasterize_forward_round",
    "gsr_rasterize_backward_two",
    "gsr_view_forward",
    "gsr_rasterize_gaussians_forward",
    "gsr_view_backward",
    "gsr_rasterize_backward_rgbd",
    "gsr_activate_forward",
    "gsr_activate_backward",
    "gsr_densify_stats",
    "gsr_densify_stats_dev",
    "gsr_refine_workspace_bytes",
    "gsr_refine_plan",
    "gsr_refine_apply",
    "gsr_adam_step",
    "gsr_rasterize_backward_det_workspace_bytes",
    "gsr_rasterize_backward_det",
    "gsr_tile_jobs_ints",
    "gsr_tile_jobs_build",
    "gsr_debug_count_staged",
    "gsr_debug_wave_trace",
    "gsr_calibrate_valu",
    "gsr_calibrate_copy",
)


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"rasterizer: native library {LIB_PATH} not found. Build it with "
            f"`make -C {CSRC}` (needs hipcc); there is no fallback implementation."
        )
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"rasterizer: {LIB_PATH} lacks symbols {missing}")
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_version.restype = C.c_int
    lib.gsr_calibrate_valu.restype = C.c_longlong
    lib.gsr_tile_jobs_ints.restype = C.c_size_t
    lib.gsr_cumsum_workspace_bytes.restype = C.c_size_t
    lib.gsr_sort_workspace_bytes.restype = C.c_size_t
    lib.gsr_reach_record_bytes.restype = C.c_size_t
    lib.gsr_depth_order_workspace_bytes.restype = C.c_size_t
    lib.gsr_bin_sorted_workspace_bytes.restype = C.c_size_t
    lib.gsr_refine_workspace_bytes.restype = C.c_size_t
    lib.gsr_tile_lists_subrange_workspace_bytes.restype = C.c_size_t
    lib.gsr_saturation_filter_workspace_bytes.restype = C.c_size_t
    lib.gsr_rasterize_forward_seg_workspace_bytes.restype = C.c_size_t
    lib.gsr_rasterize_backward_seg_workspace_bytes.restype = C.c_size_t
    lib.gsr_rasterize_backward_det_workspace_bytes.restype = C.c_size_t
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


__all__ = ["lib", "build", "LIB_PATH", "SYMBOLS"]
