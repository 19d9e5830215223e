"""The per-Gaussian activations in front of the rasterizer in one launch.

GaussianSplattingModel.get_outputs (gs_toolkit/models/vanilla_gs.py:765-826) runs

    torch.exp(scales_crop);  quats_crop / quats_crop.norm(dim=-1, keepdim=True)
    torch.sigmoid(opacities_crop)
    viewdirs = means_crop.detach() - camera_position;  viewdirs / viewdirs.norm(...)

as ~9 small launches forward and ~12 backward per iteration.  `activate_gaussians`
returns the same four tensors from one kernel (`gsr_activate_forward`) and
back-propagates through the first three with one kernel (`gsr_activate_backward`).
"""
import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from rasterizer.cuda import _call, _check, _ptr, _stream

_f32 = torch.float32


class _Activate(Function):
    @staticmethod
    def forward(ctx, means, log_scales, raw_quats, opacity_logits, campos):
        n = log_scales.shape[0]
        for t, nm in ((log_scales, "scales"), (raw_quats, "quats"), (opacity_logits, "opacities")):
            _check(t, nm, _f32)
        dev = log_scales.device
        want_dirs = campos is not None
        if want_dirs:
            _check(means, "means", _f32)
            _check(campos, "camera_position", _f32)
        with torch.cuda.device(dev):
            scales = torch.empty_like(log_scales)
            quats = torch.empty_like(raw_quats)
            opac = torch.empty_like(opacity_logits)
            dirs = torch.empty((n, 3), dtype=_f32, device=dev) if want_dirs else None
            _call("gsr_activate_forward", C.c_int(n), _ptr(means) if want_dirs else None, _ptr(log_scales),
                  _ptr(raw_quats), _ptr(opacity_logits), _ptr(campos) if want_dirs else None, _ptr(scales),
                  _ptr(quats), _ptr(opac), _ptr(dirs) if want_dirs else None, _stream(dev))
        ctx.save_for_backward(raw_quats, scales, quats, opac)
        ctx.set_materialize_grads(False)
        from rasterizer.ahead import announce_opacity

        announce_opacity(opac)  # the projection that follows may start this view's tile lists with it
        if want_dirs:
            ctx.mark_non_differentiable(dirs)
            return scales, quats, opac, dirs
        return scales, quats, opac, None

    @staticmethod
    def backward(ctx, v_scales, v_quats, v_opac, _v_dirs):
        raw_quats, scales, quats, opac = ctx.saved_tensors
        n = scales.shape[0]
        dev = scales.device
        cot = [None if t is None else _check(t.contiguous(), nm, _f32)
               for t, nm in ((v_scales, "v_scales"), (v_quats, "v_quats"), (v_opac, "v_opacities"))]
        with torch.cuda.device(dev):
            g_s = torch.empty_like(scales)
            g_q = torch.empty_like(quats)
            g_o = torch.empty_like(opac)
            _call("gsr_activate_backward", C.c_int(n), _ptr(raw_quats), _ptr(scales), _ptr(quats), _ptr(opac),
                  *[None if t is None else _ptr(t) for t in cot], _ptr(g_s), _ptr(g_q), _ptr(g_o), _stream(dev))
        return None, g_s, g_q, g_o, None


def activate_gaussians(means: Optional[Tensor], log_scales: Tensor, raw_quats: Tensor, opacity_logits: Tensor,
                       camera_position: Optional[Tensor] = None
                       ) -> Tuple[Tensor, Tensor, Tensor, Optional[Tensor]]:
    """-> (exp(log_scales) [N,3], raw_quats / |raw_quats| [N,4], sigmoid(opacity_logits) [N,1],
    normalised (means - camera_position) [N,3] or None).  Differentiable w.r.t. the three
    parameter tensors; the view directions carry no gradient (as in the reference)."""
    n = log_scales.shape[0]
    if log_scales.shape != (n, 3) or raw_quats.shape != (n, 4) or opacity_logits.numel() != n:
        raise ValueError("expected scales [N,3], quats [N,4], opacities [N,1]")
    if camera_position is not None and (means is None or means.shape != (n, 3) or camera_position.numel() != 3):
        raise ValueError("view directions need means [N,3] and a camera position [3]")
    return _Activate.apply(means.detach().contiguous() if camera_position is not None else None,
                           log_scales.contiguous(), raw_quats.contiguous(), opacity_logits.contiguous(),
                           camera_position.contiguous() if camera_position is not None else None)


@torch.no_grad()
def densify_stats_(xys_grad: Optional[Tensor], radii: Tensor, image_size: int, xys_grad_norm: Tensor,
                   vis_counts: Tensor, max_2dsize: Tensor, first: bool = False) -> None:
    """In place, for the Gaussians with ``radii > 0`` (GaussianSplattingModel.after_train,
    vanilla_gs.py:344-372): ``xys_grad_norm += |xys_grad|``, ``vis_counts += 1``,
    ``max_2dsize = max(max_2dsize, radii / image_size)`` -- one launch (``gsr_densify_stats``).
    ``first=True`` is the reference's first call after a refinement (:354-356, the
    accumulators are ``None`` there): every Gaussian starts with count 1 and its own
    gradient norm, visible or not; the accumulators need not be initialised."""
    n = radii.numel()
    _check(radii, "radii", torch.int32)
    _check(xys_grad_norm, "xys_grad_norm", _f32)
    _check(vis_counts, "vis_counts", torch.int32)
    _check(max_2dsize, "max_2dsize", _f32)
    if xys_grad is not None:
        xys_grad = _check(xys_grad.contiguous(), "xys_grad", _f32)
        if xys_grad.shape != (n, 2):
            raise ValueError("xys_grad must be [N,2]")
    if xys_grad_norm.numel() != n or vis_counts.numel() != n or max_2dsize.numel() != n:
        raise ValueError("the accumulators must have N elements")
    dev = radii.device
    with torch.cuda.device(dev):
        _call("gsr_densify_stats", C.c_int(n), _ptr(xys_grad) if xys_grad is not None else None, _ptr(radii),
              C.c_float(1.0 / float(image_size)), C.c_int(1 if first else 0), _ptr(xys_grad_norm),
              _ptr(vis_counts), _ptr(max_2dsize), _stream(dev))
