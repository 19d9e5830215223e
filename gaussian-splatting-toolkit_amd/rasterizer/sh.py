"""SH colour evaluation, differentiable in the coefficients only.

Public surface of the reference's ``rasterizer/sh.py``: ``num_sh_bases`` (:10),
``deg_from_sh`` (:22), ``spherical_harmonics`` (:36) and the autograd node behind
it (:60).  The arithmetic lives in ``csrc/sh.hip``; the view directions are
normalised inside the kernels.
"""
from torch import Tensor
from torch.autograd import Function

import rasterizer.cuda as _C

# band count per degree; the reference answers 25 for any degree above 3
_BAND_COUNT = (1, 4, 9, 16)
_DEGREE_OF = {count: degree for degree, count in enumerate(_BAND_COUNT + (25,))}


def num_sh_bases(degree: int) -> int:
    return _BAND_COUNT[degree] if 0 <= degree < len(_BAND_COUNT) else 25


def deg_from_sh(num_bases: int) -> int:
    degree = _DEGREE_OF.get(num_bases)
    assert degree is not None, "Invalid number of SH bases"
    return degree


class _SphericalHarmonics(Function):
    """inputs: (bands in use, directions [N,3], coefficients [N,K,3]) -> colours [N,3]"""

    @staticmethod
    def forward(ctx, active_degree: int, dirs: Tensor, sh: Tensor):
        ctx.sizes = (sh.shape[0], deg_from_sh(sh.shape[-2]), active_degree)
        ctx.save_for_backward(dirs)
        return _C.compute_sh_forward(*ctx.sizes, dirs, sh)

    @staticmethod
    def backward(ctx, grad_colors: Tensor):
        (dirs,) = ctx.saved_tensors
        grad_sh = _C.compute_sh_backward(*ctx.sizes, dirs, grad_colors, trusted=True)
        return None, None, grad_sh  # bands above `active_degree` receive zeros


def spherical_harmonics(degrees_to_use: int, viewdirs: Tensor, coeffs: Tensor) -> Tensor:
    """Colours [N,3] seen from ``viewdirs`` [N,3] for coefficients ``coeffs`` [N,K,3].

    Only the first ``degrees_to_use`` degrees contribute (the tensor may be sized
    for more).  No gradient reaches ``viewdirs``."""
    if coeffs.shape[-2] < num_sh_bases(degrees_to_use):
        raise AssertionError("coeffs hold fewer SH bands than degrees_to_use needs")
    return _SphericalHarmonics.apply(degrees_to_use, viewdirs.contiguous(), coeffs.contiguous())
