"""GPU: the harness trainer (toolkit call pattern: activations, SH warm-up,
L1+SSIM, retain_grad on xys, 6 Adam groups) actually learns through the HIP
rasterizer -- gradients that were merely self-consistent would not."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_short_training_run_improves_psnr():
    from harness.train import TrainConfig, train

    cfg = TrainConfig(num_gaussians=20_000, width=320, height=180, num_views=8, iters=120,
                      sh_degree=3, sh_degree_interval=30, log_every=10)
    res = train(cfg, torch.device("cuda", 0))
    assert np.isfinite(res["param_checksum"])
    assert res["losses"][-1] < 0.7 * res["losses"][0]
    assert res["psnr_end"] > res["psnr_start"] + 3.0, res


def test_ssim_matches_definition():
    from harness.train import ssim

    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.rand(64, 80, 3, device="cuda", generator=g)
    assert abs(float(ssim(a, a)) - 1.0) < 1e-5
    b = (a + 0.1 * torch.randn(64, 80, 3, device="cuda", generator=g)).clamp(0, 1)
    s = float(ssim(a, b))
    assert 0.2 < s < 0.99
    assert abs(float(ssim(b, a)) - s) < 1e-6
