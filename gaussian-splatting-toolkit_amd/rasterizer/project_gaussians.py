"""EWA projection of 3-D Gaussians to screen space (differentiable).

Mirror of the reference's ``rasterizer/project_gaussians.py`` (function
``project_gaussians`` :12-76, autograd node ``_ProjectGaussians`` :79-232):
same signature, output tuple, saved tensors and ``None`` pattern in the
backward.  The math runs in the HIP kernels of ``csrc/project.hip``.
"""
from typing import Tuple

from torch import Tensor
from torch.autograd import Function

import rasterizer.cuda as _C


def project_gaussians(
    means3d: Tensor,
    scales: Tensor,
    glob_scale: float,
    quats: Tensor,
    viewmat: Tensor,
    projmat: Tensor,
    fx: float,
    fy: float,
    cx: float,
    cy: float,
    img_height: int,
    img_width: int,
    block_width: int,
    clip_thresh: float = 0.01,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Project N Gaussians (differentiable w.r.t. ``means3d``, ``scales``, ``quats``).

    Args:
        means3d: [N,3] centres.  scales: [N,3] (already exponentiated).
        glob_scale: global scale factor.  quats: [N,4] rotations, (w,x,y,z).
        viewmat: world->camera, row-major; only the top 3x4 is read.
        projmat: full projection (P @ V), 4x4 row-major.
        fx, fy, cx, cy: pinhole intrinsics in pixels.
        img_height, img_width: output size.  block_width: tile side, 2..16.
        clip_thresh: near-plane distance; ``z <= clip_thresh`` is culled.

    Returns:
        ``(xys [N,2], depths [N], radii [N] int32, conics [N,3],
        compensation [N], num_tiles_hit [N] int32, cov3d [N,6])``.
        Culled Gaussians have ``radii == 0`` and ``num_tiles_hit == 0``.
    """
    assert block_width > 1 and block_width <= 16, "block_width must be between 2 and 16"
    return _ProjectGaussians.apply(
        means3d.contiguous(), scales.contiguous(), glob_scale, quats.contiguous(),
        viewmat.contiguous(), projmat.contiguous(), fx, fy, cx, cy, img_height, img_width,
        block_width, clip_thresh,
    )


class _ProjectGaussians(Function):
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
                img_height, img_width, block_width, clip_thresh=0.01):
        num_points = means3d.shape[-2]
        if num_points < 1 or means3d.shape[-1] != 3:
            raise ValueError(f"Invalid shape for means3d: {means3d.shape}")

        # native order puts cov3d first; the public tuple puts it last
        cov3d, xys, depths, radii, conics, compensation, num_tiles_hit = _C.project_gaussians_forward(
            num_points, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
            img_height, img_width, block_width, clip_thresh,
        )

        ctx.scalars = (num_points, glob_scale, fx, fy, cx, cy, img_height, img_width)
        # unused outputs (depths, compensation, cov3d in the RGB pass) arrive as None
        # in backward instead of N-sized zero tensors; the kernel reads None as zero
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii, num_tiles_hit)
        ctx.save_for_backward(means3d, scales, quats, viewmat, projmat, cov3d, radii, conics,
                              compensation)
        return xys, depths, radii, conics, compensation, num_tiles_hit, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_compensation, v_num_tiles_hit, v_cov3d):
        means3d, scales, quats, viewmat, projmat, cov3d, radii, conics, compensation = ctx.saved_tensors
        num_points, glob_scale, fx, fy, cx, cy, img_height, img_width = ctx.scalars
        _, _, v_mean3d, v_scale, v_quat = _C.project_gaussians_backward(
            num_points, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
            img_height, img_width, cov3d, radii, conics, compensation, v_xys, v_depths, v_conics,
            v_compensation,
        )
        # one slot per forward input: means3d, scales, glob_scale, quats, then 10 non-tensors
        return (v_mean3d, v_scale, None, v_quat) + (None,) * 10
