"""Spherical-harmonics colour evaluation (differentiable w.r.t. the coefficients).

Mirror of the reference's ``rasterizer/sh.py`` (``num_sh_bases`` :10,
``deg_from_sh`` :22, ``spherical_harmonics`` :36, ``_SphericalHarmonics`` :60).
"""
from torch import Tensor
from torch.autograd import Function

import rasterizer.cuda as _C

_BASES = {0: 1, 1: 4, 2: 9, 3: 16}
_DEGREES = {1: 0, 4: 1, 9: 2, 16: 3, 25: 4}


def num_sh_bases(degree: int) -> int:
    """(degree+1)^2 for degree 0..3, 25 for anything above (as the reference)."""
    return _BASES.get(degree, 25)


def deg_from_sh(num_bases: int) -> int:
    if num_bases in _DEGREES:
        return _DEGREES[num_bases]
    assert False, "Invalid number of SH bases"


def spherical_harmonics(degrees_to_use: int, viewdirs: Tensor, coeffs: Tensor) -> Tensor:
    """Colours [N,3] from view directions [N,3] and coefficients [N,K,3].

    ``degrees_to_use`` may be lower than the degree the coefficient tensor was
    sized for; higher bands are ignored (and get zero gradient).  Directions are
    normalised inside the kernel.  No gradient flows to ``viewdirs``.
    """
    assert coeffs.shape[-2] >= num_sh_bases(degrees_to_use)
    return _SphericalHarmonics.apply(degrees_to_use, viewdirs.contiguous(), coeffs.contiguous())


class _SphericalHarmonics(Function):
    @staticmethod
    def forward(ctx, degrees_to_use: int, viewdirs: Tensor, coeffs: Tensor):
        num_points = coeffs.shape[0]
        degree = deg_from_sh(coeffs.shape[-2])
        ctx.degrees_to_use = degrees_to_use
        ctx.degree = degree
        ctx.save_for_backward(viewdirs)
        return _C.compute_sh_forward(num_points, degree, degrees_to_use, viewdirs, coeffs)

    @staticmethod
    def backward(ctx, v_colors: Tensor):
        (viewdirs,) = ctx.saved_tensors
        v_coeffs = _C.compute_sh_backward(
            v_colors.shape[0], ctx.degree, ctx.degrees_to_use, viewdirs, v_colors
        )
        return None, None, v_coeffs
