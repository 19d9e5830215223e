"""The OPTIONAL loss terms of the reference's co-gs model -- every one of them off in its default config
(`DepthGSModelConfig`, gs_toolkit/models/depth_gs.py:93-139) -- restated as plain torch ops for the trainer harness
(`harness.train`, `model="co-gs"`).  None of this is on the rasterizer's hot path; the terms exist so that a config
that switches them on trains through the same loop.  What each follows:

  pearson_depth_loss     utils/losses.py:12-23   1 - corr(src, target), biased covariance over UNBIASED std's (as written)
  local_pearson_loss     utils/losses.py:26-45   the mean of that over int(p_corr * boxes) random box_p x box_p patches
  tv_loss                utils/losses.py:197-207 mean |d/dx| + mean |d/dy| of an [H, W] image
  scaled_log_depth_loss  depth_gs.py:492-518     edge-aware log(1 + |gt - (scale * pred + shift)|), weights exp(-|grad img|)
  scale_regularisation   depth_gs.py:450-460     0.1 * mean(max(max_scale / min_scale, ratio) - ratio), every 10th step
  sparse_loss            depth_gs.py:462-467     lambda * mean(log(o + 1e-6) + log(1 - o + 1e-6)), every 100th step

NOT restated: `depth_reg_loss` (depth_gs.py:521-528) -- it needs the Canny edge mask of the ground-truth image
(`image2canny`, utils/losses.py:48-70: OpenCV, absent from this image) -- and the planar losses (open3d RANSAC, called
from nowhere in the model).  `optional_depth_terms` raises for the former instead of guessing.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch


def pearson_depth_loss(depth_src: torch.Tensor, depth_target: torch.Tensor) -> torch.Tensor:
    """1 - Pearson correlation of two flat depth vectors (utils/losses.py:12-23).  The source divides a BIASED
    covariance (a mean) by the product of torch.std's UNBIASED deviations: for n samples the value is
    1 - (n - 1) / n * r.  Followed as written."""
    mean_src, mean_target = depth_src.mean(), depth_target.mean()
    cov = ((depth_src - mean_src) * (depth_target - mean_target)).mean()
    return 1 - cov / (depth_src.std() * depth_target.std())


def local_pearson_patches(height: int, width: int, box_p: int, p_corr: float,
                          generator: Optional[torch.Generator] = None, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """The top-left corners the source draws (utils/losses.py:28-37): int(p_corr * floor(H / box) * floor(W / box))
    patches, rows from [0, max(H - box, 0)), columns from [0, max(W - box, 0)) -- `torch.randint`'s upper bound is
    exclusive, so the last row / column is never a corner, and an image no larger than the box has no valid draw
    (the source raises there; so does this)."""
    n_corr = int(p_corr * math.floor(height / box_p) * math.floor(width / box_p))
    max_h, max_w = max(height - box_p, 0), max(width - box_p, 0)
    x0 = torch.randint(0, max_h, size=(n_corr,), device=device, generator=generator)
    y0 = torch.randint(0, max_w, size=(n_corr,), device=device, generator=generator)
    return x0, y0


def local_pearson_loss(depth_src: torch.Tensor, depth_target: torch.Tensor, box_p: int, p_corr: float,
                       generator: Optional[torch.Generator] = None,
                       corners: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """Mean of `pearson_depth_loss` over random box_p x box_p patches (utils/losses.py:26-45).  `corners`: the
    patches' top-left (row, column) indices, for a reproducible evaluation; default: drawn as the source draws them.
    One gather instead of the source's Python loop over patches: the same numbers, no per-patch launch.  A patch whose
    rendered or target depth is constant is 0 / 0 = nan, as in the source (it has no epsilon)."""
    src = depth_src.squeeze(-1) if depth_src.dim() == 3 else depth_src
    tgt = depth_target.squeeze(-1) if depth_target.dim() == 3 else depth_target
    x0, y0 = corners if corners is not None else local_pearson_patches(src.shape[0], src.shape[1], box_p, p_corr,
                                                                       generator, src.device)
    n_corr = int(x0.numel())
    if n_corr == 0:
        return (src.sum() * 0.0) / 0.0  # the source divides the empty sum by n_corr = 0: nan
    rows = x0.view(-1, 1, 1) + torch.arange(box_p, device=src.device).view(1, -1, 1)  # (full patches: corners stop
    cols = y0.view(-1, 1, 1) + torch.arange(box_p, device=src.device).view(1, 1, -1)  #  box_p short of the border)
    a = src[rows, cols].reshape(n_corr, -1)
    b = tgt[rows, cols].reshape(n_corr, -1)
    am, bm = a.mean(dim=1, keepdim=True), b.mean(dim=1, keepdim=True)
    cov = ((a - am) * (b - bm)).mean(dim=1)
    return (1 - cov / (a.std(dim=1) * b.std(dim=1))).sum() / n_corr


def tv_loss(pred: torch.Tensor) -> torch.Tensor:
    """utils/losses.py:197-207 on the [H, W] depth the model hands it (its docstring speaks of a batch; the call site,
    depth_gs.py:530-531, passes the squeezed depth image): mean |column differences| + mean |row differences|."""
    h_diff = pred[:, :-1] - pred[:, 1:]
    w_diff = pred[:-1, :] - pred[1:, :]
    return h_diff.abs().mean() + w_diff.abs().mean()


def scaled_log_depth_loss(pred_depth: torch.Tensor, gt_depth: torch.Tensor, gt_img: torch.Tensor,
                          scale: float | torch.Tensor = 1.0, shift: float | torch.Tensor = 0.0) -> torch.Tensor:
    """`log_depth` of depth_gs.py:492-518: logl1 = log(1 + |gt - (scale * pred + shift)|), weighted along x by
    exp(-mean_c |img[:, :-1] - img[:, 1:]|) and along y by the row analogue (smooth image regions count fully, edges
    less); the two weighted means are added.  pred_depth [H, W] (or [H, W, 1]), gt_img [H, W, 3]."""
    pred = pred_depth.squeeze(-1) if pred_depth.dim() == 3 else pred_depth
    logl1 = torch.log(1 + torch.abs(gt_depth - (scale * pred + shift)))
    grad_x = torch.abs(gt_img[:, :-1, :] - gt_img[:, 1:, :]).mean(-1)
    grad_y = torch.abs(gt_img[:-1, :, :] - gt_img[1:, :, :]).mean(-1)
    loss_x = torch.exp(-grad_x) * logl1[:, :-1]
    loss_y = torch.exp(-grad_y) * logl1[:-1, :]
    return loss_x.mean() + loss_y.mean()


def scale_regularisation(log_scales: torch.Tensor, max_gauss_ratio: float = 10.0) -> torch.Tensor:
    """`scale_reg` of depth_gs.py:450-460 (the model applies it on every 10th step): needle-shaped Gaussians -- longest
    over shortest axis above `max_gauss_ratio` -- pay 0.1 x the excess, averaged over all Gaussians."""
    scale_exp = torch.exp(log_scales)
    ratio = scale_exp.amax(dim=-1) / scale_exp.amin(dim=-1)
    excess = torch.maximum(ratio, torch.tensor(max_gauss_ratio, device=ratio.device, dtype=ratio.dtype)) - max_gauss_ratio
    return 0.1 * excess.mean()


def sparse_loss(opacities: torch.Tensor, sparse_lambda: float) -> torch.Tensor:
    """`sparse_loss` of depth_gs.py:462-467 (every 100th step).  The source feeds `gauss_params["opacities"]` -- the
    LOGITS, not their sigmoid -- to the two logs; followed as written (a logit outside (0, 1) makes the term nan,
    which is the source's behaviour with this switch on)."""
    return sparse_lambda * (torch.log(opacities + 1e-6) + torch.log(1 - opacities + 1e-6)).mean()


def optional_depth_terms(cfg, step: int, pred_depth: torch.Tensor, gt_depth: torch.Tensor, gt_img: torch.Tensor,
                         generator: Optional[torch.Generator] = None, mono_scale_shift=None) -> Dict[str, torch.Tensor]:
    """The `use_est_depth` branch of `DepthGSModel.get_loss_dict` (depth_gs.py:477-531) as a dict of terms the trainer
    sums unweighted (engine/trainer.py:497): local Pearson while step < depth_loss_stop_iteration, the scaled log-depth
    term when the batch carries a scale / shift, the TV term below step 20 000.  `cfg` carries the reference's field
    names (use_pearson_depth, local_patch_size, depth_loss_stop_iteration, use_scaled_est_depth,
    use_depth_regularization, using_tv_loss)."""
    terms: Dict[str, torch.Tensor] = {}
    if step < cfg.depth_loss_stop_iteration and cfg.use_pearson_depth:
        terms["depth_local_pearson"] = local_pearson_loss(pred_depth, gt_depth, cfg.local_patch_size, 0.5, generator)
    pred = pred_depth.squeeze(-1) if pred_depth.dim() == 3 else pred_depth
    if cfg.use_scaled_est_depth and mono_scale_shift is not None:
        terms["log_depth"] = scaled_log_depth_loss(pred, gt_depth, gt_img, *mono_scale_shift)
    if cfg.use_depth_regularization:
        raise NotImplementedError(
            "co-gs depth_reg_loss needs the Canny edge mask of the ground-truth image (utils/losses.py:48-70, OpenCV), "
            "which this image does not have; the term is off in the reference's default config")
    if cfg.using_tv_loss and step < 20_000:
        terms["tv_loss"] = tv_loss(pred)
    return terms
