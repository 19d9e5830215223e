"""`GaussianRasterizer` / `GaussianRasterizationSettings`: the call surface of the
Inria `diff_gaussian_rasterization` package, as a thin adapter over this
package's three ops.

BASELINE.json's north star names these classes; the reference itself does not
contain them (it calls `project_gaussians` / `spherical_harmonics` /
`rasterize_gaussians`), so this module is a
convenience for code written against the Inria API, not part of the drop-in
boundary.  It adds no kernels: everything below is argument conversion.

Conventions of the Inria API that are translated here
  * `viewmatrix` / `projmatrix` arrive TRANSPOSED (column-major: the Inria code
    stores `world_view_transform = W2C.T` and `full_proj_transform = (P @ W2C).T`);
  * intrinsics come as `tanfovx`, `tanfovy`; the principal point is the image
    centre (Inria's ndc2Pix((v+1)*S-1)/2 equals this package's 0.5*S*v + S/2 - 0.5);
  * `means2D` is a dummy [N,3] tensor whose `.grad` receives the screen-space
    gradient used for densification (x, y in its first two columns);
  * colours come either as SH coefficients `shs` [N,K,3] (evaluated towards
    `campos`, `max(sh + 0.5, 0)` like the Inria forward) or as `colors_precomp`;
  * the image is returned channels-first [3,H,W], together with `radii` [N].
Not translated (documented differences from the Inria kernels): the Inria
near-plane cull is z <= 0.2 (here: `clip_thresh`, default 0.01, as in the
reference toolkit); the Inria forward clamps alpha at 0.99 (here 0.999 forward /
0.99 backward, as the reference toolkit's kernels); `cov3D_precomp` is not
supported.
"""
from typing import NamedTuple, Optional, Tuple

import torch
from torch import Tensor

from .project_gaussians import project_gaussians
from .rasterize import rasterize_gaussians
from .sh import deg_from_sh, spherical_harmonics


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool = False
    debug: bool = False


class _ScreenGrad(torch.autograd.Function):
    """Identity on `xys` that also routes its gradient into `means2D[:, :2]`."""

    @staticmethod
    def forward(ctx, xys: Tensor, means2D: Tensor):
        return xys.view_as(xys)

    @staticmethod
    def backward(ctx, v_xys: Tensor):
        v_means2D = torch.zeros(v_xys.shape[0], 3, dtype=v_xys.dtype, device=v_xys.device)
        v_means2D[:, :2] = v_xys
        return v_xys, v_means2D


class GaussianRasterizer(torch.nn.Module):
    BLOCK_WIDTH = 16

    def __init__(self, raster_settings: GaussianRasterizationSettings, clip_thresh: float = 0.01):
        super().__init__()
        self.raster_settings = raster_settings
        self.clip_thresh = clip_thresh

    def forward(self, means3D: Tensor, means2D: Tensor, opacities: Tensor, shs: Optional[Tensor] = None,
                colors_precomp: Optional[Tensor] = None, scales: Optional[Tensor] = None,
                rotations: Optional[Tensor] = None, cov3D_precomp: Optional[Tensor] = None
                ) -> Tuple[Tensor, Tensor]:
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if cov3D_precomp is not None:
            raise NotImplementedError("cov3D_precomp is not supported by this adapter")
        if scales is None or rotations is None:
            raise Exception("Please provide scales and rotations")
        H, W = int(rs.image_height), int(rs.image_width)
        fx = W / (2.0 * rs.tanfovx)
        fy = H / (2.0 * rs.tanfovy)
        viewmat = rs.viewmatrix.t().contiguous()   # back to row-major world->camera
        projmat = rs.projmatrix.t().contiguous()   # row-major P @ V
        quats = rotations / rotations.norm(dim=-1, keepdim=True)
        xys, depths, radii, conics, comp, num_tiles_hit, _ = project_gaussians(
            means3D, scales, rs.scale_modifier, quats, viewmat[:3, :], projmat, fx, fy, W / 2.0, H / 2.0,
            H, W, self.BLOCK_WIDTH, self.clip_thresh)
        if means2D is not None and means2D.requires_grad:
            xys = _ScreenGrad.apply(xys, means2D)
        if shs is not None:
            dirs = means3D.detach() - rs.campos
            dirs = dirs / dirs.norm(dim=-1, keepdim=True)
            use = min(int(rs.sh_degree), deg_from_sh(shs.shape[-2]))
            colors = torch.clamp_min(spherical_harmonics(use, dirs, shs) + 0.5, 0.0)
        else:
            colors = colors_precomp
        opac = opacities if opacities.dim() == 2 else opacities[:, None]
        img = rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opac, H, W,
                                  self.BLOCK_WIDTH, background=rs.bg)
        return img.permute(2, 0, 1), radii
