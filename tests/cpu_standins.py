"""CPU stand-ins of the three rasterizer ops and of `refine_gaussians`, backed by the
ORACLE, for tests of host logic that must run without a GPU (the 2-rank gloo test of
the trainer).  Test infrastructure: the product has no CPU path, and nothing outside
tests/ imports this module.  Signatures follow the reference's package
(`rasterizer/project_gaussians.py:12`, `rasterizer/sh.py:34`, `rasterizer/rasterize.py:14`).
"""
import numpy as np
import torch

from oracle import oracle as O
from oracle import refine as RO

_np = lambda t: t.detach().cpu().numpy()
_t = torch.from_numpy


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W, bw, clip):
        n = means3d.shape[0]
        out = O.project_gaussians_forward(n, _np(means3d), _np(scales), glob_scale, _np(quats), _np(viewmat),
                                          _np(projmat), fx, fy, cx, cy, H, W, bw, clip)
        cov3d, xys, depths, radii, conics, comp, tiles = (_t(a) for a in out)
        ctx.args = (glob_scale, fx, fy, cx, cy, H, W)
        ctx.save_for_backward(means3d, scales, quats, viewmat, projmat, cov3d, radii, conics, comp)
        ctx.mark_non_differentiable(radii, tiles)
        return xys, depths, radii, conics, comp, tiles, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_comp, v_tiles, v_cov3d):
        means3d, scales, quats, viewmat, projmat, cov3d, radii, conics, comp = ctx.saved_tensors
        glob_scale, fx, fy, cx, cy, H, W = ctx.args
        n = means3d.shape[0]
        z = lambda v, like: np.zeros(like.shape, np.float32) if v is None else _np(v)
        _, _, v_mean, v_scale, v_quat = O.project_gaussians_backward(
            n, _np(means3d), _np(scales), glob_scale, _np(quats), _np(viewmat), _np(projmat), fx, fy, cx, cy, H, W,
            _np(cov3d), _np(radii), _np(conics), _np(comp), z(v_xys, conics[:, :2]), z(v_depths, comp),
            z(v_conics, conics), z(v_comp, comp))
        return (_t(v_mean), _t(v_scale), None, _t(v_quat)) + (None,) * 10


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, img_height, img_width,
                      block_width, clip_thresh=0.01):
    return _Project.apply(means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, img_height,
                          img_width, block_width, clip_thresh)


class _SH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degrees_to_use, viewdirs, coeffs):
        n, K = coeffs.shape[0], coeffs.shape[1]
        degree = {1: 0, 4: 1, 9: 2, 16: 3, 25: 4}[K]
        ctx.args = (n, degree, degrees_to_use)
        ctx.save_for_backward(viewdirs)
        return _t(O.compute_sh_forward(n, degree, degrees_to_use, _np(viewdirs), _np(coeffs)))

    @staticmethod
    def backward(ctx, v_colors):
        n, degree, use = ctx.args
        (viewdirs,) = ctx.saved_tensors
        return None, None, _t(O.compute_sh_backward(n, degree, use, _np(viewdirs), _np(v_colors.contiguous())))


def spherical_harmonics(degrees_to_use, viewdirs, coeffs):
    return _SH.apply(degrees_to_use, viewdirs, coeffs)


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, H, W, bw, background, return_alpha):
        n = xys.shape[0]
        tb = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
        I, cum = O.compute_cumulative_intersects(_np(num_tiles_hit))
        ctx.dims = (H, W, bw, I)
        if I < 1:
            img = torch.ones(H, W, colors.shape[-1]) * background
            Ts = torch.zeros(H, W)
            ctx.save_for_backward(xys, conics, colors, opacity)
            return (img, 1 - Ts) if return_alpha else img
        _, _, _, ids, bins = O.bin_and_sort_gaussians(n, I, _np(xys), _np(depths), _np(radii), cum, tb, bw)
        img, Ts, idx = O.rasterize_forward(tb, (bw, bw, 1), (W, H, 1), ids, bins, _np(xys), _np(conics),
                                           _np(colors), _np(opacity), _np(background))
        ctx.lists = (ids, bins, Ts, idx, _np(background))
        ctx.save_for_backward(xys, conics, colors, opacity)
        return (_t(img), 1 - _t(Ts)) if return_alpha else _t(img)

    @staticmethod
    def backward(ctx, v_img, v_alpha=None):
        xys, conics, colors, opacity = ctx.saved_tensors
        H, W, bw, I = ctx.dims
        if I < 1:
            return (torch.zeros_like(xys), None, None, torch.zeros_like(conics), None, torch.zeros_like(colors),
                    torch.zeros_like(opacity)) + (None,) * 5
        ids, bins, Ts, idx, bg = ctx.lists
        v_img = np.zeros((H, W, colors.shape[-1]), np.float32) if v_img is None else _np(v_img.contiguous())
        v_alpha = np.zeros((H, W), np.float32) if v_alpha is None else _np(v_alpha.contiguous())
        vxy, vconic, vcol, vop = O.rasterize_backward(H, W, bw, ids, bins, _np(xys), _np(conics), _np(colors),
                                                      _np(opacity), bg, Ts, idx, v_img, v_alpha)
        return (_t(vxy), None, None, _t(vconic), None, _t(vcol), _t(vop).reshape(opacity.shape)) + (None,) * 5


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                        block_width, background=None, return_alpha=False):
    if background is None:
        background = torch.ones(colors.shape[-1])
    return _Rasterize.apply(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                            block_width, background, return_alpha)


def refine_gaussians(params, moments, stats, cfg, step, num_train_data, max_dim, samples=None, seed=0):
    """`gs_fused.refine_gaussians` on CPU tensors through the numpy oracle; same return
    convention (the input tensors themselves when nothing changes)."""
    ocfg = RO.RefineConfig(**{f: getattr(cfg, f) for f in RO.RefineConfig.__dataclass_fields__})
    p_in = {k: _np(v) for k, v in params.items()}
    m_in = None if moments is None else {k: (_np(a), _np(b)) for k, (a, b) in moments.items()}
    s_in = None if stats is None else tuple(_np(a).astype(np.float32) for a in stats)
    p, m, info = RO.refine(p_in, m_in, s_in, ocfg, step, num_train_data, max_dim, samples=samples, seed=seed)
    n_in, n_out = p_in["means"].shape[0], p["means"].shape[0]
    moved = info["culls"] is not None and (n_out != n_in or bool(info["culls"].any()))
    out_info = {"n_in": n_in, "n_out": n_out, "opacity_reset": info["opacity_reset"]}
    if not moved:
        if info["opacity_reset"]:
            with torch.no_grad():
                params["opacities"].copy_(_t(p["opacities"]))
                if moments is not None and "opacities" in moments:
                    for t in moments["opacities"]:
                        t.zero_()
        return dict(params), (None if moments is None else dict(moments)), out_info
    new_p = {k: _t(np.ascontiguousarray(v)) for k, v in p.items()}
    new_m = None if m is None else {k: (_t(np.ascontiguousarray(a)), _t(np.ascontiguousarray(b)))
                                    for k, (a, b) in m.items()}
    return new_p, new_m, out_info
