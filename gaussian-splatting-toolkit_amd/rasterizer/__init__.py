"""``rasterizer`` -- differentiable Gaussian-splatting rasterizer for AMD
Instinct MI355X (gfx950).

Drop-in for the package of the same name that Gaussian-Splatting-Toolkit builds
from ``gs_toolkit/gs_components`` (a gsplat 0.1.x fork): identical modules,
functions, signatures and tensor conventions, so
``gs_toolkit.models.{vanilla_gs,depth_gs,surface_gs}`` import and train
unchanged on PyTorch-ROCm.  The device code is hand-written HIP
(``../csrc/*.hip``) behind a C ABI (``include/gsraster.h``).
"""
import warnings
from typing import Any

import torch

from .project_gaussians import project_gaussians
from .rasterize import rasterize_gaussians
from .sh import spherical_harmonics
from .utils import (
    bin_and_sort_gaussians,
    compute_cov2d_bounds,
    compute_cumulative_intersects,
    get_tile_bin_edges,
    map_gaussian_to_intersects,
)
from .version import __version__


def _deprecated_function(class_name: str, target, target_name: str):
    """The reference keeps a ``torch.autograd.Function`` per public function for
    backwards compatibility (rasterizer/__init__.py:43-166): ``forward`` warns
    and forwards, ``backward`` is not implemented."""

    def forward(ctx, *args, **kwargs):
        warnings.warn(f"{class_name} is deprecated, use {target_name} instead", DeprecationWarning)
        return target(*args, **kwargs)

    def backward(ctx: Any, *grad_outputs: Any) -> Any:
        raise NotImplementedError

    return type(
        class_name,
        (torch.autograd.Function,),
        {"forward": staticmethod(forward), "backward": staticmethod(backward), "__module__": __name__},
    )


MapGaussiansToIntersects = _deprecated_function(
    "MapGaussiansToIntersects", map_gaussian_to_intersects, "map_gaussian_to_intersects")
ComputeCumulativeIntersects = _deprecated_function(
    "ComputeCumulativeIntersects", compute_cumulative_intersects, "compute_cumulative_intersects")
ComputeCov2dBounds = _deprecated_function(
    "ComputeCov2dBounds", compute_cov2d_bounds, "compute_cov2d_bounds")
GetTileBinEdges = _deprecated_function("GetTileBinEdges", get_tile_bin_edges, "get_tile_bin_edges")
BinAndSortGaussians = _deprecated_function(
    "BinAndSortGaussians", bin_and_sort_gaussians, "bin_and_sort_gaussians")
ProjectGaussians = _deprecated_function("ProjectGaussians", project_gaussians, "project_gaussians")
RasterizeGaussians = _deprecated_function(
    "RasterizeGaussians", rasterize_gaussians, "rasterize_gaussians")
NDRasterizeGaussians = _deprecated_function(
    "NDRasterizeGaussians", rasterize_gaussians, "rasterize_gaussians")
SphericalHarmonics = _deprecated_function(
    "SphericalHarmonics", spherical_harmonics, "spherical_harmonics")

__all__ = [
    "__version__",
    "project_gaussians",
    "rasterize_gaussians",
    "spherical_harmonics",
    "bin_and_sort_gaussians",
    "compute_cumulative_intersects",
    "compute_cov2d_bounds",
    "get_tile_bin_edges",
    "map_gaussian_to_intersects",
    "ProjectGaussians",
    "RasterizeGaussians",
    "BinAndSortGaussians",
    "ComputeCumulativeIntersects",
    "ComputeCov2dBounds",
    "GetTileBinEdges",
    "MapGaussiansToIntersects",
    "SphericalHarmonics",
    "NDRasterizeGaussians",
]
