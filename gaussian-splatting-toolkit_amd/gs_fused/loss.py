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
WORKSPACE_DOUBLES = 2 * 64  # GSR_LOSS_WORKSPACE_DOUBLES (include/gsraster.h)


class _L1SSIM(Function):
    @staticmethod
    def forward(ctx, pred: Tensor, gt: Tensor, ssim_lambda: float, clamp_pred: bool):
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
            work = torch.empty((WORKSPACE_DOUBLES,), dtype=torch.float64, device=dev)
            loss = torch.empty((), dtype=_f32, device=dev)
            terms = torch.empty((2,), dtype=_f32, device=dev)  # L1 mean, SSIM mean
            _call("gsr_l1_ssim_forward", C.c_uint(H), C.c_uint(W), C.c_float(ssim_lambda),
                  C.c_int(1 if clamp_pred else 0), _ptr(pred), _ptr(gt), _ptr(maps), _ptr(work), _ptr(loss),
                  _ptr(terms), _stream(dev))
        ctx.save_for_backward(pred, gt, maps)
        ctx.ssim_lambda = float(ssim_lambda)
        ctx.clamp_pred = bool(clamp_pred)
        ctx.hw = (H, W)
        ctx.mark_non_differentiable(terms)
        return loss, terms

    @staticmethod
    def backward(ctx, v_loss, _v_terms):
        pred, gt, maps = ctx.saved_tensors
        H, W = ctx.hw
        dev = pred.device
        up = v_loss.to(_f32).reshape(1).contiguous()
        with torch.cuda.device(dev):
            v_pred = torch.empty_like(pred)
            _call("gsr_l1_ssim_backward", C.c_uint(H), C.c_uint(W), C.c_float(ctx.ssim_lambda),
                  C.c_int(1 if ctx.clamp_pred else 0), _ptr(up), _ptr(pred), _ptr(gt), _ptr(maps), _ptr(v_pred),
                  _stream(dev))
        return v_pred, None, None, None


def l1_ssim_loss(pred: Tensor, gt: Tensor, ssim_lambda: float = 0.2, return_terms: bool = False,
                 clamp_pred: bool = False):
    """Scalar loss (fp32, on device).  With `return_terms`, also the L1 mean and the
    SSIM value (detached diagnostics; gradients flow through the loss only).
    `clamp_pred`: compute the loss of `torch.clamp(pred, max=1.0)` (what the models
    feed it, vanilla_gs.py:857) without that op and its backward."""
    loss, terms = _L1SSIM.apply(pred, gt, ssim_lambda, clamp_pred)
    if return_terms:
        return loss, terms[0], terms[1]
    return loss


class L1SSIMLoss(torch.nn.Module):
    def __init__(self, ssim_lambda: float = 0.2, clamp_pred: bool = False):
        super().__init__()
        self.ssim_lambda = ssim_lambda
        self.clamp_pred = clamp_pred

    def forward(self, pred: Tensor, gt: Tensor) -> Tensor:
        return l1_ssim_loss(pred, gt, self.ssim_lambda, clamp_pred=self.clamp_pred)


class _L1(Function):
    @staticmethod
    def forward(ctx, pred: Tensor, gt: Tensor, weight: float, clamp_pred: bool):
        if pred.shape != gt.shape or pred.numel() == 0:
            raise ValueError(f"expected two images of one (non-empty) shape, got {tuple(pred.shape)} and {tuple(gt.shape)}")
        pred = _check(pred.contiguous(), "pred", _f32)
        gt = _check(gt.contiguous(), "gt", _f32)
        dev = pred.device
        with torch.cuda.device(dev):
            work = torch.empty((64,), dtype=torch.float64, device=dev)
            loss = torch.empty((), dtype=_f32, device=dev)
            _call("gsr_l1_forward", C.c_longlong(pred.numel()), C.c_float(weight), C.c_int(1 if clamp_pred else 0),
                  _ptr(pred), _ptr(gt), _ptr(work), _ptr(loss), _stream(dev))
        ctx.save_for_backward(pred, gt)
        ctx.weight, ctx.clamp_pred = float(weight), bool(clamp_pred)
        return loss

    @staticmethod
    def backward(ctx, v_loss):
        pred, gt = ctx.saved_tensors
        dev = pred.device
        up = v_loss.to(_f32).reshape(1).contiguous()
        with torch.cuda.device(dev):
            v_pred = torch.empty_like(pred)
            _call("gsr_l1_backward", C.c_longlong(pred.numel()), C.c_float(ctx.weight),
                  C.c_int(1 if ctx.clamp_pred else 0), _ptr(up), _ptr(pred), _ptr(gt), _ptr(v_pred), _stream(dev))
        return v_pred, None, None, None


def l1_loss(pred: Tensor, gt: Tensor, weight: float = 1.0, clamp_pred: bool = False) -> Tensor:
    """``weight * |gt - pred|.mean()`` in one streaming kernel each way -- the photometric loss of the co-gs
    model AS ITS SOURCE COMPUTES IT: `DepthGSModel.get_loss_dict` (depth_gs.py:445-448) writes
    ``loss_dict["main_loss"] = (1 - ssim_lambda) * Ll1`` and puts ``+ssim_lambda * simloss`` on a line of its own,
    an expression statement whose value is dropped, so ``weight = 1 - ssim_lambda`` and no SSIM term.
    `clamp_pred`: the loss of ``torch.clamp(pred, max=1.0)`` (depth_gs.py:343) without that op."""
    return _L1.apply(pred, gt, weight, clamp_pred)


class _DepthL1(Function):
    @staticmethod
    def forward(ctx, depth: Tensor, alpha: Tensor, gt: Tensor):
        if depth.numel() != alpha.numel() or depth.numel() != gt.numel() or depth.numel() == 0:
            raise ValueError("depth, alpha and gt_depth must have the same (non-zero) number of pixels")
        d = _check(depth.contiguous(), "depth", _f32)
        a = _check(alpha.contiguous(), "alpha", _f32)
        g = _check(gt.contiguous(), "gt_depth", _f32)
        dev = d.device
        with torch.cuda.device(dev):
            far = d.detach().max().reshape(1)  # depth_im.detach().max() of the reference
            work = torch.empty((64,), dtype=torch.float64, device=dev)
            loss = torch.empty((), dtype=_f32, device=dev)
            _call("gsr_depth_l1_forward", C.c_longlong(d.numel()), _ptr(d), _ptr(a), _ptr(g), _ptr(far),
                  _ptr(work), _ptr(loss), _stream(dev))
        ctx.save_for_backward(d, a, g, far)
        ctx.shapes = (depth.shape, alpha.shape)
        return loss

    @staticmethod
    def backward(ctx, v_loss):
        d, a, g, far = ctx.saved_tensors
        dev = d.device
        up = v_loss.to(_f32).reshape(1).contiguous()
        with torch.cuda.device(dev):
            v_d = torch.empty_like(d)
            v_a = torch.empty_like(a)
            _call("gsr_depth_l1_backward", C.c_longlong(d.numel()), _ptr(up), _ptr(d), _ptr(a), _ptr(g),
                  _ptr(far), _ptr(v_d), _ptr(v_a), _stream(dev))
        return v_d.view(ctx.shapes[0]), v_a.view(ctx.shapes[1]), None


def depth_l1_loss(depth_acc: Tensor, alpha: Tensor, gt_depth: Tensor) -> Tensor:
    """Depth head of the co-gs model in two launches (+ one max): with
    ``pred = where(alpha > 0, depth_acc / alpha, depth_acc.detach().max())``
    (depth_gs.py:356-363) returns ``|gt * (gt > 0) - pred * (gt > 0)|.mean()``
    (depth_gs.py:531-538).  ``depth_acc`` is the raw output of the depth compositing
    pass, ``alpha`` the accumulated opacity of the RGB pass; differentiable w.r.t. both."""
    return _DepthL1.apply(depth_acc, alpha, gt_depth)
