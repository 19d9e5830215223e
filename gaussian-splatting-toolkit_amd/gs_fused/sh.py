"""SH colours from split coefficients, without the `torch.cat`.

The models hold the DC band and the higher bands as two parameters and join them
for every render (gs_toolkit/models/vanilla_gs.py:809):

    colors_crop = torch.cat((features_dc_crop[:, None, :], features_rest_crop), dim=1)
    rgbs = spherical_harmonics(n, viewdirs, colors_crop)

At SH degree 3 the cat is a 192-MB copy per forward and autograd splits the
gradient back with two strided copies per backward -- more HBM traffic than the
SH kernels themselves.  `spherical_harmonics_split(n, viewdirs, features_dc,
features_rest)` returns the same colours and writes the two gradients in place
(`gsr_sh_forward_split` / `gsr_sh_backward_split`, include/gsraster.h).
"""
import ctypes as C

import torch
from torch import Tensor
from torch.autograd import Function

from rasterizer.cuda import _call, _check, _ptr, _stream
from rasterizer.sh import deg_from_sh, num_sh_bases, spherical_harmonics

_f32 = torch.float32


class _SplitSH(Function):
    @staticmethod
    def forward(ctx, degrees_to_use: int, viewdirs: Tensor, dc: Tensor, rest: Tensor, shift: float,
                clamp_zero: bool):
        n = dc.shape[0]
        degree = deg_from_sh(rest.shape[-2] + 1)
        for t, nm in ((viewdirs, "viewdirs"), (dc, "features_dc"), (rest, "features_rest")):
            _check(t, nm, _f32)
        dev = dc.device
        with torch.cuda.device(dev):
            colors = torch.empty((n, 3), dtype=_f32, device=dev)
            _call("gsr_sh_forward_split", C.c_uint(n), C.c_uint(degree), C.c_uint(degrees_to_use),
                  _ptr(viewdirs), _ptr(dc), _ptr(rest), _ptr(colors), C.c_float(shift),
                  C.c_int(1 if clamp_zero else 0), _stream(dev))
        ctx.degree, ctx.degrees_to_use = degree, degrees_to_use
        ctx.shapes = (dc.shape, rest.shape)
        ctx.clamped = bool(clamp_zero)
        ctx.save_for_backward(viewdirs, colors if clamp_zero else None)
        return colors

    @staticmethod
    def backward(ctx, v_colors: Tensor):
        viewdirs, colors = ctx.saved_tensors
        n = viewdirs.shape[0]
        v_colors = _check(v_colors.contiguous(), "v_colors", _f32)
        dev = viewdirs.device
        with torch.cuda.device(dev):
            v_dc = torch.empty(ctx.shapes[0], dtype=_f32, device=dev)
            v_rest = torch.empty(ctx.shapes[1], dtype=_f32, device=dev)
            _call("gsr_sh_backward_split", C.c_uint(n), C.c_uint(ctx.degree), C.c_uint(ctx.degrees_to_use),
                  _ptr(viewdirs), _ptr(v_colors), _ptr(colors) if ctx.clamped else None, _ptr(v_dc),
                  _ptr(v_rest), _stream(dev))
        return None, None, v_dc, v_rest, None, None


def spherical_harmonics_split(degrees_to_use: int, viewdirs: Tensor, features_dc: Tensor,
                              features_rest: Tensor, shift: float = 0.0, clamp_zero: bool = False) -> Tensor:
    """Colours [N,3] from `features_dc` [N,3] (or [N,1,3]) and `features_rest`
    [N,K-1,3]; equal to `spherical_harmonics(degrees_to_use, viewdirs,
    torch.cat((features_dc[:, None], features_rest), 1))`, differentiable w.r.t.
    both coefficient tensors.  K - 1 must be 3, 8 or 15 (degree 1..3); a model
    without higher bands (K = 1) or with degree 4 takes the concatenating path.
    ``shift`` / ``clamp_zero``: the models' epilogue ``torch.clamp(rgbs + 0.5, min=0.0)``
    (vanilla_gs.py:826) inside the same kernels: ``shift=0.5, clamp_zero=True``."""
    if features_dc.dim() == 3:
        if features_dc.shape[1] != 1:
            raise ValueError("features_dc must be [N,3] or [N,1,3]")
    elif features_dc.dim() != 2:
        raise ValueError("features_dc must be [N,3] or [N,1,3]")
    if features_rest.dim() != 3 or features_rest.shape[0] != features_dc.shape[0] or features_rest.shape[2] != 3 \
            or features_dc.shape[-1] != 3 or viewdirs.shape != (features_dc.shape[0], 3):
        raise ValueError("expected viewdirs [N,3], features_dc [N,3], features_rest [N,K-1,3]")
    K = features_rest.shape[1] + 1
    assert K >= num_sh_bases(degrees_to_use)
    if K not in (4, 9, 16):
        dc3 = features_dc if features_dc.dim() == 3 else features_dc[:, None, :]
        out = spherical_harmonics(degrees_to_use, viewdirs, torch.cat((dc3, features_rest), dim=1))
        if shift != 0.0:
            out = out + shift
        return torch.clamp(out, min=0.0) if clamp_zero else out
    return _SplitSH.apply(degrees_to_use, viewdirs.contiguous(), features_dc.contiguous(),
                          features_rest.contiguous(), float(shift), bool(clamp_zero))


def sh_backward_views(degree: int, degrees_to_use: int, means3d: Tensor, camera_positions: Tensor, v_colors: Tensor,
                      scale: float = 1.0, split: bool = True):
    """``gsr_sh_backward_views``: the SH gradient summed over several views, from the views' colour cotangents:
    ``scale * sum_r B(normalize(means3d - camera_positions[r])) (x) v_colors[r]`` -> ``(v_dc [N,3], v_rest [N,K-1,3])``
    (``split``) or ``v_coeffs [N,K,3]``.  ``camera_positions`` [V,3] and ``v_colors`` [V,N,3] may be strided over
    the views (rows of one gathered message).  What `harness.parallel.GradientExchange` forms after all-gathering the
    ranks' 12-byte colour cotangents instead of all-reducing 12 K bytes of SH gradient per Gaussian."""
    _check(means3d, "means3d", _f32)
    n, V = means3d.shape[0], v_colors.shape[0]
    K = num_sh_bases(degree)
    for t, nm, inner in ((camera_positions, "camera_positions", 3), (v_colors, "v_colors", 3 * n)):
        if not (t.is_cuda and t.dtype == _f32 and t.shape[0] == V and t[0].numel() == inner
                and (V == 0 or t[0].is_contiguous())):
            raise RuntimeError(f"sh_backward_views: {nm} must be float32 CUDA, [V, ...] with contiguous views")
    dev = means3d.device
    with torch.cuda.device(dev):
        v_dc = v_rest = v_coeffs = None
        if split:
            v_dc = torch.empty((n, 3), dtype=_f32, device=dev)
            v_rest = torch.empty((n, K - 1, 3), dtype=_f32, device=dev)
        else:
            v_coeffs = torch.empty((n, K, 3), dtype=_f32, device=dev)
        _call("gsr_sh_backward_views", C.c_uint(n), C.c_uint(degree), C.c_uint(degrees_to_use), C.c_uint(V),
              _ptr(means3d), _ptr(camera_positions), C.c_size_t(camera_positions.stride(0) if V > 1 else 3),
              _ptr(v_colors), C.c_size_t(v_colors.stride(0) if V > 1 else 3 * n), C.c_float(scale),
              _ptr(v_dc) if split else None, _ptr(v_rest) if split and K > 1 else None,
              _ptr(v_coeffs) if not split else None, _stream(dev))
    return (v_dc, v_rest) if split else v_coeffs
