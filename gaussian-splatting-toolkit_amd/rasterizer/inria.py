"""`GaussianRasterizer` / `GaussianRasterizationSettings`: the call surface of the
Inria `diff_gaussian_rasterization` package, as a thin adapter over this
package's three ops.

BASELINE.json's north star names these classes; the reference itself does not
contain them (it calls `project_gaussians` / `spherical_harmonics` /
`rasterize_gaussians`), so this module is a
convenience for code written against the Inria API, not part of the drop-in
boundary.  It adds no kernels: everything below is argument conversion (the
precomputed-covariance projection is the same kernel with `scales` / `quats` absent,
include/gsraster.h).

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
  * `cov3D_precomp` [N,6] (upper triangle xx xy xz yy yz zz, already scaled) may
    replace `scales` + `rotations`, and receives its own gradient;
  * forks of the Inria rasterizer also return a depth and an alpha image:
    `forward(..., return_depth=True, return_alpha=True)` appends `depth` [1,H,W]
    (sum_i z_i alpha_i T_i, from the same compositing pass) and `alpha` [1,H,W].
Not translated (documented differences from the Inria kernels): the Inria
near-plane cull is z <= 0.2 (here: `clip_thresh`, default 0.01, as in the
reference toolkit); the Inria forward clamps alpha at 0.99 (here 0.999 forward /
0.99 backward, as the reference toolkit's kernels).
"""
from typing import NamedTuple, Optional, Tuple

import torch
from torch import Tensor

import rasterizer.cuda as _C

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


class _ProjectPrecomputed(torch.autograd.Function):
    """`project_gaussians` with the 3-D covariances handed in (Inria's `cov3D_precomp`):
    differentiable w.r.t. `means3d` and `cov3d`."""

    @staticmethod
    def forward(ctx, means3d, cov3d, viewmat, projmat, fx, fy, cx, cy, img_height, img_width, block_width,
                clip_thresh):
        n = means3d.shape[0]
        camera = (viewmat, projmat, fx, fy, cx, cy, img_height, img_width)
        _, xys, depths, radii, conics, comp, tiles = _C.project_gaussians_forward(
            n, means3d, None, 1.0, None, *camera, block_width, clip_thresh, cov3d_precomp=cov3d)
        ctx.static = (n,) + camera[2:]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii, tiles)
        ctx.save_for_backward(means3d, viewmat, projmat, cov3d, radii, conics, comp)
        return xys, depths, radii, conics, comp, tiles

    @staticmethod
    def backward(ctx, g_xys, g_depths, _g_radii, g_conics, g_comp, _g_tiles):
        means3d, viewmat, projmat, cov3d, radii, conics, comp = ctx.saved_tensors
        n, fx, fy, cx, cy, H, W = ctx.static
        _, v_cov3d, v_mean3d, _, _ = _C.project_gaussians_backward(
            n, means3d, None, 1.0, None, viewmat, projmat, fx, fy, cx, cy, H, W, cov3d, radii, conics, comp,
            g_xys, g_depths, g_conics, g_comp)
        return (v_mean3d, v_cov3d) + (None,) * 10


class GaussianRasterizer(torch.nn.Module):
    BLOCK_WIDTH = 16

    def __init__(self, raster_settings: GaussianRasterizationSettings, clip_thresh: float = 0.01):
        super().__init__()
        self.raster_settings = raster_settings
        self.clip_thresh = clip_thresh

    def forward(self, means3D: Tensor, means2D: Tensor, opacities: Tensor, shs: Optional[Tensor] = None,
                colors_precomp: Optional[Tensor] = None, scales: Optional[Tensor] = None,
                rotations: Optional[Tensor] = None, cov3D_precomp: Optional[Tensor] = None,
                return_depth: bool = False, return_alpha: bool = False):
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        H, W = int(rs.image_height), int(rs.image_width)
        fx = W / (2.0 * rs.tanfovx)
        fy = H / (2.0 * rs.tanfovy)
        viewmat = rs.viewmatrix.t().contiguous()   # back to row-major world->camera
        projmat = rs.projmatrix.t().contiguous()   # row-major P @ V
        if cov3D_precomp is not None:
            xys, depths, radii, conics, comp, num_tiles_hit = _ProjectPrecomputed.apply(
                means3D.contiguous(), cov3D_precomp.contiguous(), viewmat[:3, :].contiguous(), projmat, fx, fy,
                W / 2.0, H / 2.0, H, W, self.BLOCK_WIDTH, self.clip_thresh)
        else:
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
        if return_depth:
            from gs_fused import rasterize_gaussians_rgbd  # RGB + depth from one compositing pass

            img, alpha, depth = rasterize_gaussians_rgbd(xys, depths, radii, conics, num_tiles_hit, colors, depths,
                                                         opac, H, W, background=rs.bg)
            out = (img.permute(2, 0, 1), radii, depth.permute(2, 0, 1))
            return out + ((alpha[None],) if return_alpha else ())
        if return_alpha:
            img, alpha = rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opac, H, W,
                                             self.BLOCK_WIDTH, background=rs.bg, return_alpha=True)
            return img.permute(2, 0, 1), radii, alpha[None]
        img = rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opac, H, W,
                                  self.BLOCK_WIDTH, background=rs.bg)
        return img.permute(2, 0, 1), radii
