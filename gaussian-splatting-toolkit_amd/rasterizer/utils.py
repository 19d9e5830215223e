"""Tile binning helpers (none of them differentiable).

Mirror of the reference's ``rasterizer/utils.py``:
``map_gaussian_to_intersects`` :12, ``get_tile_bin_edges`` :55,
``compute_cov2d_bounds`` :84, ``compute_cumulative_intersects`` :106,
``bin_and_sort_gaussians`` :128 -- same names, arguments and returned tuples.
Differences in mechanism: the scan and the sort run in rocPRIM behind the C ABI
(``gsr_cumsum_tiles``, ``gsr_sort_intersects``) instead of ``torch.cumsum`` /
``torch.sort`` + ``torch.gather``; the sort is stable and covers only the
significant key bits.
"""
from typing import Tuple

from torch import Tensor

import rasterizer.cuda as _C


def map_gaussian_to_intersects(
    num_points: int, num_intersects: int, xys: Tensor, depths: Tensor, radii: Tensor,
    cum_tiles_hit: Tensor, tile_bounds: Tuple[int, int, int], block_size: int,
) -> Tuple[Tensor, Tensor]:
    """One ``(tile_id << 32 | depth_bits)`` key and one Gaussian id per
    (Gaussian, covered tile) pair, written at the offsets ``cum_tiles_hit`` gives.

    Returns ``(isect_ids int64 [I], gaussian_ids int32 [I])``.
    """
    return _C.map_gaussian_to_intersects(
        num_points, num_intersects, xys.contiguous(), depths.contiguous(), radii.contiguous(),
        cum_tiles_hit.contiguous(), tile_bounds, block_size,
    )


def get_tile_bin_edges(num_intersects: int, isect_ids_sorted: Tensor,
                       tile_bounds: Tuple[int, int, int]) -> Tensor:
    """``tile_bins[t] = (first, one_past_last)`` index into the sorted
    intersection list for tile ``t``; ``(0, 0)`` for tiles nothing touches."""
    return _C.get_tile_bin_edges(num_intersects, isect_ids_sorted.contiguous(), tile_bounds)


def compute_cov2d_bounds(cov2d: Tensor) -> Tuple[Tensor, Tensor]:
    """Conic (inverse covariance, upper triangle) and 3-sigma radius from the
    upper triangle of a 2-D covariance, ``cov2d`` [N,3] -> ([N,3], [N,1])."""
    assert cov2d.shape[-1] == 3, (
        f"Expected input cov2d to be of shape (*batch, 3) (upper triangular values), "
        f"but got {tuple(cov2d.shape)}"
    )
    num_pts = cov2d.shape[0]
    assert num_pts > 0
    return _C.compute_cov2d_bounds(num_pts, cov2d.contiguous())


def compute_cumulative_intersects(num_tiles_hit: Tensor) -> Tuple[int, Tensor]:
    """Inclusive scan of the per-Gaussian tile counts.

    Returns ``(num_intersects, cum_tiles_hit int32 [N])``.  Reading the total
    back is the one host sync of the pipeline (the reference's ``.item()``,
    utils.py:124); the buffers of the following stages are sized by it.
    """
    cum_tiles_hit = _C.cumsum_tiles(num_tiles_hit.contiguous())
    num_intersects = int(cum_tiles_hit[-1].item())
    return num_intersects, cum_tiles_hit


def bin_and_sort_gaussians(
    num_points: int, num_intersects: int, xys: Tensor, depths: Tensor, radii: Tensor,
    cum_tiles_hit: Tensor, tile_bounds: Tuple[int, int, int], block_size: int,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Emit, sort and bin the tile intersections.

    Returns ``(isect_ids_unsorted, gaussian_ids_unsorted, isect_ids_sorted,
    gaussian_ids_sorted, tile_bins)`` -- unsorted arrays included, as in the
    reference, so tests can inspect them.
    """
    isect_ids, gaussian_ids = map_gaussian_to_intersects(
        num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds, block_size
    )
    num_tiles = int(tile_bounds[0]) * int(tile_bounds[1])
    isect_ids_sorted, gaussian_ids_sorted = _C.sort_intersects(isect_ids, gaussian_ids, num_tiles)
    tile_bins = get_tile_bin_edges(num_intersects, isect_ids_sorted, tile_bounds)
    return isect_ids, gaussian_ids, isect_ids_sorted, gaussian_ids_sorted, tile_bins
