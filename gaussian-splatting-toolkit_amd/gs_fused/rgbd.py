"""RGB and a depth image from ONE compositing pass.

When a depth image is wanted the models composite twice per view
(gs_toolkit/models/vanilla_gs.py:822-855, depth_gs.py:330-363):

    rgb, alpha = rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, rgbs, opacities, H, W, B,
                                     background=background, return_alpha=True)
    depth_im = rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, depths[:, None].repeat(1, 3),
                                   opacities, H, W, B, background=torch.zeros(3))[..., 0:1]

i.e. two full forward and two full backward passes over the same sorted lists (the
list building is shared through the rasterizer's cache, the compositing is not).
`rasterize_gaussians_rgbd` composites a fourth channel in the same pass:

    rgb, alpha, depth_im = rasterize_gaussians_rgbd(xys, depths, radii, conics, num_tiles_hit, rgbs, depths,
                                                    opacities, H, W, background=background)

Same values (the RGB image is bit-identical, the extra image equals channel 0 of the
second pass) and the same gradients; block width 16 only.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

import rasterizer.cuda as _C

_f32 = torch.float32
BLOCK = 16


class _RasterizeRGBD(Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, extra, opacity, img_height, img_width,
                background, extra_background):
        from rasterizer.rasterize import build_tile_lists

        n = xys.size(0)
        tile_bounds = ((img_width + BLOCK - 1) // BLOCK, (img_height + BLOCK - 1) // BLOCK, 1)
        dev = xys.device

        from rasterizer import rasterize as _R

        # alpha = 1 - T and the backward's cleared accumulators come out of the compositing launch
        acc = None
        if any(ctx.needs_input_grad[i] for i in (0, 3, 5, 6, 7)) and not _R.is_deterministic():
            acc = _C.backward_accumulators(n, 4, dev)
        alpha_out = [None]

        def composite(ids, bins):
            img_, ext_, Ts_, idx_, alpha_out[0] = _C.rasterize_forward_rgbd(
                tile_bounds, (img_width, img_height, 1), ids, bins, xys, conics, colors, extra, opacity, background,
                extra_background, want_alpha=True, zero=acc)
            return img_, ext_, Ts_, idx_

        # deep scenes: lists in two segments, compositing in two rounds (rasterizer/rasterize.py, "two-round lists")
        state = []

        def round1(ids_, bins1, flags):
            with torch.cuda.device(dev):
                state[:] = [torch.empty((img_height, img_width, 3), dtype=torch.float32, device=dev),
                            torch.empty((img_height, img_width), dtype=torch.float32, device=dev),
                            torch.empty((img_height, img_width), dtype=torch.float32, device=dev),
                            torch.empty((img_height, img_width), dtype=torch.int32, device=dev), flags]
                alpha_out[0] = torch.empty((img_height, img_width), dtype=torch.float32, device=dev)
            _C.rasterize_forward_round(1, tile_bounds, (img_width, img_height, 1), ids_, bins1, 0, xys, conics, colors,
                                       extra, opacity, background, extra_background, state[0], state[1], state[2],
                                       state[3], flags, out_alpha=alpha_out[0], zero=acc)

        def round2(ids_, aux):
            _C.rasterize_forward_round(2, tile_bounds, (img_width, img_height, 1), ids_, aux[1], aux[2], xys, conics,
                                       colors, extra, opacity, background, extra_background, state[0], state[1], state[2],
                                       state[3], state[4], out_alpha=alpha_out[0])
            return state[0], state[1], state[2], state[3]

        def both_rounds(ids_, bins1, aux):
            flags = torch.zeros((tile_bounds[0] * tile_bounds[1],), dtype=torch.int32, device=dev)
            round1(ids_, bins1, flags)
            return round2(ids_, aux)

        is_two = lambda a: isinstance(a, tuple) and len(a) == 3 and a[0] == "two"
        num_intersects, ids, bins, finish = build_tile_lists(
            xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width, BLOCK, round1=round1)
        two_aux = _R.last_list_aux() if is_two(_R.last_list_aux()) else None
        if finish is not None:  # lists sized from the previous view: composite, then check the count
            if two_aux is not None:
                img, ext, Ts, idx = round2(ids, two_aux) if state else both_rounds(ids, bins, two_aux)
            else:
                img, ext, Ts, idx = composite(ids, bins)
            num_intersects, ids, bins, rebuilt = finish()
            if rebuilt:
                two_aux = _R.last_list_aux() if is_two(_R.last_list_aux()) else None
                img, ext, Ts, idx = both_rounds(ids, bins, two_aux) if two_aux is not None else composite(ids, bins)
        elif num_intersects >= 1:
            img, ext, Ts, idx = both_rounds(ids, bins, two_aux) if two_aux is not None else composite(ids, bins)
        if num_intersects < 1:
            img = torch.ones(img_height, img_width, 3, device=dev) * background
            ext = torch.full((img_height, img_width), float(extra_background), device=dev)
            # final_Ts = 0, i.e. alpha = 1, for a view with nothing on screen: the quirk of
            # `rasterize_gaussians(return_alpha=True)` (rasterize.py:119-127), kept so that the
            # fused and the two-pass route agree (tests/test_gpu_api.py::test_empty_view_*)
            Ts = torch.zeros(img_height, img_width, device=dev)
            idx = torch.zeros(img_height, img_width, dtype=torch.int32, device=dev)
            ids = torch.zeros(0, dtype=torch.int32, device=dev)
            bins = torch.zeros(0, 2, dtype=torch.int32, device=dev)
            acc, alpha_out[0] = None, None
        ctx.accumulators = acc
        ctx.det = _R.last_list_aux() if (_R.is_deterministic() and num_intersects >= 1) else None
        ctx.two = (two_aux[1], two_aux[2]) if (two_aux is not None and num_intersects >= 1) else None
        ctx.set_materialize_grads(False)
        ctx.meta = (img_height, img_width, num_intersects, float(extra_background))
        ctx.save_for_backward(ids, bins, xys, conics, colors, extra, opacity, background, Ts, idx)
        return img, (alpha_out[0] if alpha_out[0] is not None else 1 - Ts), ext

    @staticmethod
    def backward(ctx, v_img, v_alpha, v_ext):
        ids, bins, xys, conics, colors, extra, opacity, background, Ts, idx = ctx.saved_tensors
        H, W, num_intersects, ebg = ctx.meta
        n = xys.size(0)
        dev = xys.device
        if num_intersects < 1:
            z = torch.zeros_like
            return (z(xys), None, None, z(conics), None, z(colors), z(extra), z(opacity)) + (None,) * 4
        v_img = torch.zeros(H, W, 3, device=dev) if v_img is None else v_img
        v_ext = torch.zeros(H, W, device=dev) if v_ext is None else v_ext
        if ctx.two is not None:
            acc, ctx.accumulators = ctx.accumulators, None
            v_xy, v_conic, v_colors, v_extra, v_opacity = _C.rasterize_backward_two(
                H, W, ids, bins, ctx.two[0], ctx.two[1], xys, conics, colors, extra, opacity, background, ebg, Ts, idx,
                v_img, v_ext, v_alpha, accumulators=acc)
        elif ctx.det is not None:  # fixed summation order (rasterizer.rasterize.set_deterministic)
            v_xy, v_conic, v_colors, v_extra, v_opacity = _C.rasterize_backward_det(
                H, W, ids, bins, xys, conics, colors, opacity, background, Ts, idx, v_img, v_alpha, *ctx.det,
                extra=extra, extra_background=ebg, v_output_extra=v_ext)
        else:
            acc, ctx.accumulators = ctx.accumulators, None  # a second backward (retain_graph) clears its own
            v_xy, v_conic, v_colors, v_extra, v_opacity = _C.rasterize_backward_rgbd(
                H, W, ids, bins, xys, conics, colors, extra, opacity, background, ebg, Ts, idx, v_img, v_ext,
                v_alpha, accumulators=acc)
        return (v_xy, None, None, v_conic, None, v_colors, v_extra.view(extra.shape),
                v_opacity.reshape(opacity.shape)) + (None,) * 4


def rasterize_gaussians_rgbd(xys: Tensor, depths: Tensor, radii: Tensor, conics: Tensor, num_tiles_hit: Tensor,
                             colors: Tensor, extra: Tensor, opacity: Tensor, img_height: int, img_width: int,
                             background: Optional[Tensor] = None, extra_background: float = 0.0
                             ) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (rgb [H,W,3], alpha [H,W], extra image [H,W,1]): `rasterize_gaussians(..., colors,
    return_alpha=True)` and the channel-0 image of `rasterize_gaussians(..., extra repeated
    as 3 colours, background=extra_background)` from one pass.  ``extra``: [N] or [N,1]
    (typically ``depths``).  Differentiable w.r.t. xys, conics, colors, extra, opacity."""
    if colors.dim() != 2 or colors.shape[1] != 3:
        raise ValueError("colors must have dimensions (N, 3)")
    if extra.numel() != xys.shape[0]:
        raise ValueError("extra must hold one scalar per Gaussian")
    if background is None:
        background = torch.ones(3, dtype=_f32, device=colors.device)
    for t, nm in ((xys, "xys"), (conics, "conics"), (colors, "colors"), (extra, "extra"), (opacity, "opacity"),
                  (background, "background")):
        if not t.is_cuda:
            raise RuntimeError(f"{nm} must be a CUDA tensor")
    img, alpha, ext = _RasterizeRGBD.apply(
        xys.contiguous(), depths.contiguous(), radii.contiguous(), conics.contiguous(), num_tiles_hit.contiguous(),
        colors.contiguous().float(), extra.contiguous().float(), opacity.contiguous(), int(img_height),
        int(img_width), background.contiguous().float(), float(extra_background))
    return img, alpha, ext[..., None]
