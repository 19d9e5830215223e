"""EWA projection of 3-D Gaussians to screen space (differentiable).

Same call surface as the reference's ``rasterizer/project_gaussians.py``
(``project_gaussians`` :12-76, autograd node :79-232): argument order, the 7-tuple
that comes back, and which inputs receive gradients.  The arithmetic is in
``csrc/project.hip``.
"""
from typing import Tuple

from torch import Tensor
from torch.autograd import Function

import rasterizer.cuda as _C


def _ahead(*args):
    from rasterizer.ahead import speculate_lists  # (imported late: the package is still being imported)

    speculate_lists(*args)


_Out = Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]


class _ProjectGaussians(Function):
    """forward(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
    block_width, clip_thresh) -> (xys, depths, radii, conics, compensation, num_tiles_hit, cov3d)"""

    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
                img_height, img_width, block_width, clip_thresh=0.01):
        n = means3d.shape[-2]
        if n < 1 or means3d.shape[-1] != 3:
            raise ValueError(f"Invalid shape for means3d: {means3d.shape}")
        camera = (viewmat, projmat, fx, fy, cx, cy, img_height, img_width)
        # the native tuple starts with cov3d, the public one ends with it
        cov3d, xys, depths, radii, conics, compensation, num_tiles_hit = _C.project_gaussians_forward(
            n, means3d, scales, glob_scale, quats, *camera, block_width, clip_thresh)
        # everything this view's tile lists depend on exists now: start building them on the side stream, next to
        # whatever the caller does before it calls rasterize_gaussians (rasterize.py, "lists built ahead of time")
        _ahead(xys, depths, radii, conics, num_tiles_hit, img_height, img_width, block_width)
        ctx.static = (n, glob_scale) + camera[2:]
        # Outputs nobody differentiates through (depths, compensation, cov3d in an RGB
        # pass) reach backward() as None rather than as N-sized zero tensors; the
        # kernel treats a missing cotangent as zero.
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii, num_tiles_hit)
        ctx.save_for_backward(means3d, scales, quats, viewmat, projmat, cov3d, radii, conics, compensation)
        return xys, depths, radii, conics, compensation, num_tiles_hit, cov3d

    @staticmethod
    def backward(ctx, g_xys, g_depths, _g_radii, g_conics, g_compensation, _g_tiles, _g_cov3d):
        means3d, scales, quats, viewmat, projmat, cov3d, radii, conics, compensation = ctx.saved_tensors
        n, glob_scale, fx, fy, cx, cy, img_height, img_width = ctx.static
        grads = _C.project_gaussians_backward(
            n, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, img_height, img_width,
            cov3d, radii, conics, compensation, g_xys, g_depths, g_conics, g_compensation, trusted=True)
        g_means, g_scales, g_quats = grads[2:]  # (v_cov2d, v_cov3d) come first and stay internal
        # slots: means3d, scales, glob_scale, quats, then the ten camera / image arguments
        return (g_means, g_scales, None, g_quats) + (None,) * 10


def project_gaussians(means3d: Tensor, scales: Tensor, glob_scale: float, quats: Tensor, viewmat: Tensor,
                      projmat: Tensor, fx: float, fy: float, cx: float, cy: float, img_height: int,
                      img_width: int, block_width: int, clip_thresh: float = 0.01) -> _Out:
    """Project N Gaussians; differentiable w.r.t. ``means3d``, ``scales`` and ``quats``.

    means3d [N,3]; scales [N,3] (already exponentiated) times ``glob_scale``; quats
    [N,4] as (w,x,y,z); viewmat world->camera (row-major, top 3x4 used); projmat the
    full 4x4 ``P @ V``; fx, fy, cx, cy in pixels; tiles of ``block_width`` (2..16)
    pixels; Gaussians with camera ``z <= clip_thresh`` are culled.

    Returns ``(xys [N,2], depths [N], radii [N] i32, conics [N,3], compensation [N],
    num_tiles_hit [N] i32, cov3d [N,6])``; culled Gaussians have ``radii == 0`` and
    ``num_tiles_hit == 0`` and zeros elsewhere."""
    assert block_width > 1 and block_width <= 16, "block_width must be between 2 and 16"
    tensors = [t.contiguous() for t in (means3d, scales, quats, viewmat, projmat)]
    return _ProjectGaussians.apply(tensors[0], tensors[1], glob_scale, tensors[2], tensors[3], tensors[4],
                                   fx, fy, cx, cy, img_height, img_width, block_width, clip_thresh)
