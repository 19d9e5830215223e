"""The render call the toolkit's models make, reproduced outside the toolkit.

``render_view`` follows ``GaussianSplattingModel.get_outputs``
(gs_toolkit/models/vanilla_gs.py:765-855) / ``DepthGSModel.get_outputs``
(gs_toolkit/models/depth_gs.py:225-363) step by step, using only the public
``rasterizer`` API: project -> (retain xys grad) -> SH -> clamp(+0.5) ->
rasterize(return_alpha) -> optional second rasterisation of depths.
gs_toolkit itself cannot be imported in this image (tyro, jaxtyping, viser,
open3d ... are absent), so the contract is restated here for tests and bench.
"""
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from rasterizer.project_gaussians import project_gaussians
from rasterizer.rasterize import rasterize_gaussians
from rasterizer.sh import spherical_harmonics

BLOCK_WIDTH = 16  # vanilla_gs.py:762-764


@dataclass
class CameraTensors:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    viewmat: torch.Tensor  # [4,4]
    projmat: torch.Tensor  # [4,4] = P @ V
    campos: torch.Tensor  # [3]

    # [fx, fy, cx, cy, width, height] on the device: the toolkit's `Cameras` keeps its intrinsics there and
    # `get_outputs` reads them back one `.item()` at a time (vanilla_gs.py:736-740,772-773) -- `caller_syncs="camera"`
    scalars: Optional[torch.Tensor] = None

    @staticmethod
    def from_numpy(cam, device) -> "CameraTensors":
        t = lambda a: torch.from_numpy(a).to(device)
        return CameraTensors(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                             t(cam.viewmat), t(cam.projmat), t(cam.campos),
                             torch.tensor([cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height], dtype=torch.float32,
                                          device=device))


def render_view(
    means3d: torch.Tensor,      # [N,3]
    scales: torch.Tensor,       # [N,3] already exp()'d
    quats: torch.Tensor,        # [N,4] already normalised
    opacities: torch.Tensor,    # [N,1] already sigmoid()'d
    sh_coeffs,                  # [N,K,3] (features_dc ++ features_rest), or the pair
                                # (features_dc [N,3], features_rest [N,K-1,3]): no torch.cat (gs_fused)
    cam: CameraTensors,
    background: torch.Tensor,   # [3]
    sh_degree_to_use: int,
    rasterize_mode: str = "classic",
    render_depth: bool = False,
    retain_xys_grad: bool = False,
    clamp_rgb: bool = True,
    viewdirs: Optional[torch.Tensor] = None,  # normalised means3d - campos, if the caller already has them
    fused_depth: bool = False,  # RGB and depth image from one compositing pass (gs_fused.rasterize_gaussians_rgbd)
    sh_exchange=None,  # (parallel.GradientExchange, names, leaves) with names / leaves = ("features_dc",
                       # "features_rest") / those parameters, or ("sh_coeffs",) / ([N,K,3],): data parallel, the SH
                       # gradient is formed from the ranks' gathered colour cotangents (GradientExchange, `sh_views`)
    normalise_depth: bool = True,  # False: "depth" stays None and the caller takes the accumulated depth ("depth_acc")
                                   # and the alpha to gs_fused.depth_l1_loss, which normalises inside its kernels
    caller_syncs=False,  # True: block the host where the UNCHANGED models do -- `if (self.radii).sum() == 0`
                         # (vanilla_gs.py:784) and `assert (num_tiles_hit > 0).any()` (:811); "camera": also the
                         # intrinsics read back from the device ahead of the projection (:736-740, :772-773).
                         # False: the same ops with no read-back -- what a caller that WAS changed could do
) -> Dict[str, Optional[torch.Tensor]]:
    H, W = cam.height, cam.width
    if caller_syncs == "camera" and cam.scalars is not None:
        # cx.item(), cy.item(), math.atan(width / (2 fx)), math.atan(height / (2 fy)), width.item(), height.item(),
        # fx.item(), fy.item(): eight read-backs, the first of which drains the stream (the previous iteration's
        # optimizer step included)
        sc = cam.scalars
        _ = (sc[2].item(), sc[3].item(), float(sc[4] / (2 * sc[0])), float(sc[5] / (2 * sc[1])), sc[4].item(),
             sc[5].item(), sc[0].item(), sc[1].item())
    xys, depths, radii, conics, comp, num_tiles_hit, cov3d = project_gaussians(
        means3d, scales, 1, quats, cam.viewmat[:3, :], cam.projmat, cam.fx, cam.fy, cam.cx, cam.cy,
        H, W, BLOCK_WIDTH,
    )
    if caller_syncs and (radii).sum() == 0:  # vanilla_gs.py:784-794: nothing on screen, the background
        rgb = background.repeat(H, W, 1)
        # (`depth_acc`: the accumulated, un-normalised depth the fused depth loss takes together with the alpha -- zeros
        #  over a zero alpha, so that loss trains on the reference's constant image for this view instead of meeting a
        #  None)
        return {"rgb": rgb, "alpha": background.new_zeros(H, W, 1), "depth": background.new_ones(H, W, 1) * 10,
                "depth_acc": background.new_zeros(H, W, 1), "xys": xys, "radii": radii, "depths": depths, "conics": conics,
                "num_tiles_hit": num_tiles_hit, "rgbs": None}
    if retain_xys_grad and xys.requires_grad:
        xys.retain_grad()  # densification reads xys.grad (vanilla_gs.py:352-353,797-798)

    if viewdirs is None:
        viewdirs = means3d.detach() - cam.campos
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    split = isinstance(sh_coeffs, (tuple, list))
    if split:
        from gs_fused import spherical_harmonics_split

        # `torch.clamp(rgbs + 0.5, min=0.0)` inside the SH kernels
        sh_fn = lambda: spherical_harmonics_split(sh_degree_to_use, viewdirs, sh_coeffs[0], sh_coeffs[1], shift=0.5,  # noqa: E731
                                                  clamp_zero=True)
    else:
        sh_fn = lambda: spherical_harmonics(sh_degree_to_use, viewdirs, sh_coeffs)  # noqa: E731
    rgbs = None
    if sh_exchange is not None:
        exchange, names, leaves = sh_exchange
        K = leaves[-1].shape[1] + (1 if len(leaves) == 2 else 0)
        rgbs = exchange.deferred_sh_colors(sh_fn, names, leaves, means3d, cam.campos, {1: 0, 4: 1, 9: 2, 16: 3}[K],
                                           sh_degree_to_use, clamped=split)
    if rgbs is None:
        rgbs = sh_fn()
    if not split:
        rgbs = torch.clamp(rgbs + 0.5, min=0.0)

    if caller_syncs:
        assert (num_tiles_hit > 0).any()  # vanilla_gs.py:811 (a second blocking read-back)

    if rasterize_mode == "antialiased":
        opac = opacities * comp[:, None]
    elif rasterize_mode == "classic":
        opac = opacities
    else:
        raise ValueError("Unknown rasterize_mode: %s" % rasterize_mode)

    depth_im = depth_acc = None
    if render_depth and fused_depth:
        from gs_fused import rasterize_gaussians_rgbd

        rgb, alpha, depth_im = rasterize_gaussians_rgbd(xys, depths, radii, conics, num_tiles_hit, rgbs, depths, opac,
                                                        H, W, background=background)
    else:
        rgb, alpha = rasterize_gaussians(
            xys, depths, radii, conics, num_tiles_hit, rgbs, opac, H, W, BLOCK_WIDTH,
            background=background, return_alpha=True,
        )
    alpha = alpha[..., None]
    if clamp_rgb:
        rgb = torch.clamp(rgb, max=1.0)
    if render_depth:
        if depth_im is None:
            depth_im = rasterize_gaussians(
                xys, depths, radii, conics, num_tiles_hit, depths[:, None].repeat(1, 3), opac, H, W,
                BLOCK_WIDTH, background=torch.zeros(3, device=means3d.device),
            )[..., 0:1]
        depth_acc = depth_im
        depth_im = torch.where(alpha > 0, depth_im / alpha, depth_im.detach().max()) if normalise_depth else None
    return {"rgb": rgb, "alpha": alpha, "depth": depth_im, "depth_acc": depth_acc, "xys": xys, "radii": radii,
            "depths": depths, "conics": conics, "num_tiles_hit": num_tiles_hit, "rgbs": rgbs}
