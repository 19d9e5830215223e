"""(1 - lambda) * L1 + lambda * (1 - SSIM) in two HIP kernels.

Drop-in for the loss of `GaussianSplattingModel.get_loss_dict`
(gs_toolkit/models/vanilla_gs.py:926-944):

    Ll1 = torch.abs(gt_img - pred_img).mean()
    simloss = 1 - self.ssim(gt_img.permute(2, 0, 1)[None], pred_img.permute(2, 0, 1)[None])
    main_loss = (1 - ssim_lambda) * Ll1 + ssim_lambda * simloss

with `self.ssim = pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)`.
`l1_ssim_loss(pred_img, gt_img, ssim_lambda)` returns the same scalar; its backward
writes d loss / d pred_img -- the cotangent the compositing backward consumes --
in one kernel.  Differentiable w.r.t. `pred` only (the ground truth is data).
"""
import ctypes as C

import torch
from torch import Tensor
from torch.autograd import Function

from rasterizer.cuda import _call, _check, _ptr, _stream

_f32 = torch.float32
SUM_SLOTS = 64  # GSR_LOSS_SUM_SLOTS (include/gsraster.h)


class _L1SSIM(Function):
    @staticmethod
    def forward(ctx, pred: Tensor, gt: Tensor, ssim_lambda: float):
        if pred.dim() != 3 or pred.shape[-1] != 3 or pred.shape != gt.shape:
            raise ValueError(f"expected two [H,W,3] images, got {tuple(pred.shape)} and {tuple(gt.shape)}")
        H, W = int(pred.shape[0]), int(pred.shape[1])
        if H <= 10 or W <= 10:
            raise ValueError("images must be larger than the 11x11 SSIM window")
        pred = _check(pred.contiguous(), "pred", _f32)
        gt = _check(gt.contiguous(), "gt", _f32)
        dev = pred.device
        with torch.cuda.device(dev):
            maps = torch.empty((9, H - 10, W - 10), dtype=_f32, device=dev)
            sums = torch.empty((2, SUM_SLOTS), dtype=torch.float64, device=dev)
            _call("gsr_l1_ssim_forward", C.c_uint(H), C.c_uint(W), _ptr(pred), _ptr(gt), _ptr(maps),
                  _ptr(sums), _stream(dev))
        tot = sums.sum(dim=1)  # the kernel spreads its atomics over SUM_SLOTS partial sums
        l1 = tot[0] / (3.0 * H * W)
        ssim = tot[1] / (3.0 * (H - 10) * (W - 10))
        loss = ((1.0 - ssim_lambda) * l1 + ssim_lambda * (1.0 - ssim)).to(_f32)
        ctx.save_for_backward(pred, gt, maps)
        ctx.ssim_lambda = float(ssim_lambda)
        ctx.hw = (H, W)
        l1, ssim = l1.to(_f32), ssim.to(_f32)
        ctx.mark_non_differentiable(l1, ssim)
        return loss, l1, ssim

    @staticmethod
    def backward(ctx, v_loss, v_l1, v_ssim):
        pred, gt, maps = ctx.saved_tensors
        H, W = ctx.hw
        dev = pred.device
        up = v_loss.to(_f32).reshape(1).contiguous()
        with torch.cuda.device(dev):
            v_pred = torch.empty_like(pred)
            _call("gsr_l1_ssim_backward", C.c_uint(H), C.c_uint(W), C.c_float(ctx.ssim_lambda), _ptr(up),
                  _ptr(pred), _ptr(gt), _ptr(maps), _ptr(v_pred), _stream(dev))
        return v_pred, None, None


def l1_ssim_loss(pred: Tensor, gt: Tensor, ssim_lambda: float = 0.2, return_terms: bool = False):
    """Scalar loss (fp32, on device).  With `return_terms`, also the L1 mean and the
    SSIM value (detached diagnostics; gradients flow through the loss only)."""
    loss, l1, ssim = _L1SSIM.apply(pred, gt, ssim_lambda)
    if return_terms:
        return loss, l1.detach(), ssim.detach()
    return loss


class L1SSIMLoss(torch.nn.Module):
    def __init__(self, ssim_lambda: float = 0.2):
        super().__init__()
        self.ssim_lambda = ssim_lambda

    def forward(self, pred: Tensor, gt: Tensor) -> Tensor:
        return l1_ssim_loss(pred, gt, self.ssim_lambda)
