#!/usr/bin/env python3
"""Golden vectors for the loss head (row f2).

The reference computes `(1-l)*|gt-pred|.mean() + l*(1 - SSIM(gt, pred))` with
`pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)`
(gs_toolkit/models/vanilla_gs.py:181,926-944; pytorch_msssim pinned "1.0.0" in
pyproject.toml:27).  That package is neither vendored in the reference nor
installed here, so its published algorithm is restated below with the same torch
primitives it uses (separable `F.conv2d` with groups=C, valid padding, 11-tap
Gaussian of sigma 1.5, K = (0.01, 0.03)); gradients come from torch.autograd.

    python tests/golden/make_golden_loss.py
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))


def _fspecial_gauss_1d(size, sigma):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return (g / g.sum())[None, None, None]  # [1,1,1,size] like pytorch_msssim


def gaussian_filter(x, win):
    C = x.shape[1]
    out = x
    for i, s in enumerate(x.shape[2:]):
        out = F.conv2d(out, weight=win.repeat(C, 1, 1, 1).transpose(2 + i, -1), stride=1, padding=0, groups=C)
    return out


def ssim_msssim(X, Y, data_range=1.0, K=(0.01, 0.03)):
    win = _fspecial_gauss_1d(11, 1.5).to(X.dtype)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = gaussian_filter(X, win), gaussian_filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = gaussian_filter(X * X, win) - mu1_sq
    sigma2_sq = gaussian_filter(Y * Y, win) - mu2_sq
    sigma12 = gaussian_filter(X * Y, win) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1).mean()  # per channel, then size_average


def main():
    out = {}
    for name, (H, W), seed, lam in (("a", (24, 37), 0, 0.2), ("b", (48, 33), 1, 0.2), ("c", (11 + 5, 64), 2, 0.5)):
        g = torch.Generator().manual_seed(seed)
        gt = torch.rand(H, W, 3, generator=g).double()
        pred = (gt + 0.15 * torch.randn(H, W, 3, generator=g).double()).clamp(0, 1)
        pred[2:5, 3:9] = gt[2:5, 3:9]  # exact matches: sign(0) = 0 in the L1 term
        pred.requires_grad_(True)
        l1 = (gt - pred).abs().mean()
        ss = ssim_msssim(gt.permute(2, 0, 1)[None], pred.permute(2, 0, 1)[None])
        loss = (1 - lam) * l1 + lam * (1 - ss)
        loss.backward()
        out.update({f"{name}_pred": pred.detach().float().numpy(), f"{name}_gt": gt.float().numpy(),
                    f"{name}_lambda": np.float32(lam), f"{name}_loss": np.float64(loss.item()),
                    f"{name}_l1": np.float64(l1.item()), f"{name}_ssim": np.float64(ss.item()),
                    f"{name}_grad": pred.grad.float().numpy()})
        print(name, H, W, float(loss))
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **out)


if __name__ == "__main__":
    main()
