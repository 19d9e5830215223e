"""Loss head: oracle vs golden vectors (CPU); HIP kernels vs
oracle and vs golden (GPU)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

CASES = ["a", "b", "c"]


def load(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "loss.npz")))


@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_golden(golden_dir, name):
    g = load(golden_dir)
    loss, l1, ss, v = O.l1_ssim_loss(g[f"{name}_pred"], g[f"{name}_gt"], float(g[f"{name}_lambda"]))
    # goldens were computed in float64 from float64 inputs; the fixtures store float32 inputs
    assert abs(l1 - float(g[f"{name}_l1"])) < 1e-6
    assert abs(ss - float(g[f"{name}_ssim"])) < 1e-5
    assert abs(loss - float(g[f"{name}_loss"])) < 1e-5
    ref = g[f"{name}_grad"]
    # an element with pred == gt in one precision but not the other flips the sign term
    stable = np.abs(g[f"{name}_pred"].astype(np.float64) - g[f"{name}_gt"]) > 1e-6
    assert np.abs(v - ref)[stable].max() < 1e-3 * np.abs(ref).max()
    assert np.linalg.norm((v - ref)[stable]) / np.linalg.norm(ref[stable]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_vs_golden_and_oracle(golden_dir, name):
    import torch

    from gs_fused import l1_ssim_loss

    g = load(golden_dir)
    lam = float(g[f"{name}_lambda"])
    pred = torch.from_numpy(g[f"{name}_pred"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(g[f"{name}_gt"]).cuda()
    loss, l1, ss = l1_ssim_loss(pred, gt, lam, return_terms=True)
    assert abs(float(loss) - float(g[f"{name}_loss"])) < 1e-5
    assert abs(float(l1) - float(g[f"{name}_l1"])) < 1e-6 and abs(float(ss) - float(g[f"{name}_ssim"])) < 1e-5
    (3.0 * loss).backward()
    _, _, _, v = O.l1_ssim_loss(g[f"{name}_pred"], g[f"{name}_gt"], lam)
    got = pred.grad.cpu().numpy() / 3.0
    assert np.abs(got - v).max() < 1e-7 + 1e-4 * np.abs(v).max()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(1080, 1920), (237, 401), (12, 12)])
def test_hip_vs_oracle_sizes(H, W):
    import torch

    from gs_fused import L1SSIMLoss

    rng = np.random.default_rng(H + W)
    gt = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    pred = np.clip(gt + 0.1 * rng.standard_normal((H, W, 3)), 0, 1).astype(np.float32)
    p = torch.from_numpy(pred).cuda().requires_grad_(True)
    loss = L1SSIMLoss(0.2)(p, torch.from_numpy(gt).cuda())
    loss.backward()
    ref_loss, _, _, v = O.l1_ssim_loss(pred, gt, 0.2)
    assert abs(float(loss) - ref_loss) < 2e-6
    got = p.grad.cpu().numpy()
    assert np.abs(got - v).max() < 1e-4 * np.abs(v).max()
    assert np.linalg.norm(got - v) / np.linalg.norm(v) < 1e-5


@pytest.mark.gpu
def test_loss_errors():
    import torch

    from gs_fused import l1_ssim_loss

    with pytest.raises(ValueError):
        l1_ssim_loss(torch.zeros(8, 8, 3, device="cuda"), torch.zeros(8, 8, 3, device="cuda"))
    with pytest.raises(ValueError):
        l1_ssim_loss(torch.zeros(20, 20, 3, device="cuda"), torch.zeros(20, 21, 3, device="cuda"))
    with pytest.raises(RuntimeError):
        l1_ssim_loss(torch.zeros(20, 20, 3), torch.zeros(20, 20, 3))


@pytest.mark.gpu
def test_clamp_pred_equals_torch_clamp():
    """l1_ssim_loss(pred, gt, clamp_pred=True) == l1_ssim_loss(torch.clamp(pred, max=1), gt),
    value and gradient (zero where pred > 1)."""
    import torch

    from gs_fused import l1_ssim_loss

    rng = np.random.default_rng(5)
    H, W = 97, 131
    gt = torch.from_numpy(rng.uniform(0, 1, (H, W, 3)).astype(np.float32)).cuda()
    base = torch.from_numpy(rng.uniform(0, 1.4, (H, W, 3)).astype(np.float32)).cuda()
    a = base.clone().requires_grad_(True)
    b = base.clone().requires_grad_(True)
    la, l1a, sa = l1_ssim_loss(a, gt, 0.2, return_terms=True, clamp_pred=True)
    lb, l1b, sb = l1_ssim_loss(torch.clamp(b, max=1.0), gt, 0.2, return_terms=True)
    assert abs(float(la) - float(lb)) < 1e-6 and abs(float(l1a) - float(l1b)) < 1e-6 and abs(float(sa) - float(sb)) < 1e-6
    la.backward()
    lb.backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-9)
    assert float(a.grad[base > 1].abs().max()) == 0.0 and int((base > 1).sum()) > 1000


@pytest.mark.gpu
def test_depth_l1_head_equals_torch_ops():
    """gs_fused.depth_l1_loss == the co-gs depth normalisation + masked L1 (depth_gs.py:356-363,
    531-538), value and both gradients."""
    import torch

    from gs_fused import depth_l1_loss

    rng = np.random.default_rng(9)
    H, W = 120, 200
    alpha = np.clip(rng.uniform(-0.3, 1.0, (H, W, 1)), 0, 1).astype(np.float32)      # ~25 % uncovered
    depth = (alpha * rng.uniform(1, 9, (H, W, 1))).astype(np.float32)
    gt = (rng.uniform(0.5, 10, (H, W)) * (rng.uniform(0, 1, (H, W)) > 0.2)).astype(np.float32)  # holes = 0
    d1, a1 = (torch.from_numpy(x).cuda().requires_grad_(True) for x in (depth, alpha))
    d2, a2 = (torch.from_numpy(x).cuda().requires_grad_(True) for x in (depth, alpha))
    g = torch.from_numpy(gt).cuda()
    mine = depth_l1_loss(d1, a1, g)
    pred = torch.where(a2 > 0, d2 / a2, d2.detach().max()).squeeze(-1)
    nz = g > 0
    ref = torch.abs(g * nz - pred * nz).mean()
    assert abs(float(mine) - float(ref)) < 1e-6 * max(1.0, float(ref))
    (2.5 * mine).backward()
    (2.5 * ref).backward()
    # where alpha == 0 the torch ops give 0 * inf = NaN (the division's backward under a
    # masked-out `where`); those pixels have no splats, so the value is never used -- here: 0
    cov = a2.detach() > 0
    assert torch.isnan(d2.grad[~cov]).all() and int((~cov).sum()) > 1000
    assert float(d1.grad[~cov].abs().max()) == 0.0 and float(a1.grad[~cov].abs().max()) == 0.0
    assert torch.allclose(d1.grad[cov], d2.grad[cov], rtol=1e-5, atol=1e-9)
    assert torch.allclose(a1.grad[cov], a2.grad[cov], rtol=1e-5, atol=1e-8)
    with pytest.raises(ValueError):
        depth_l1_loss(d1, a1, g[:-1])
