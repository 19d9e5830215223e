"""Row f1: the one-launch Adam step.  The oracle (numpy restatement of
torch.optim.Adam's update) is pinned on torch.optim.Adam itself (CPU, here);
the HIP kernel is checked against the oracle and against torch's optimiser on
the GPU, through `gs_fused.FusedAdam` -> `gsr_adam_step` (C ABI)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

# learning rates / eps of the toolkit (configs/method_configs.py:47-80)
LRS = [1.6e-4, 0.0025, 0.0025 / 20, 0.05, 0.005, 0.001]
EPS = 1e-15


def make(shapes, seed):
    rng = np.random.default_rng(seed)
    ps = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    gs = [[(rng.standard_normal(s) * 10.0 ** rng.uniform(-6, 0)).astype(np.float32) for s in shapes] for _ in range(6)]
    return ps, gs


def oracle_run(ps, gs, lrs, eps):
    ps = [p.copy() for p in ps]
    ms = [np.zeros_like(p) for p in ps]
    vs = [np.zeros_like(p) for p in ps]
    for step, g in enumerate(gs, 1):
        for i in range(len(ps)):
            ps[i], ms[i], vs[i] = O.adam_step(ps[i], g[i], ms[i], vs[i], step, lrs[i], eps=eps)
    return ps, ms, vs


def close(a, b, rel=2e-6):
    return np.abs(a - b).max() <= rel * max(1e-30, np.abs(b).max())


def test_oracle_is_torch_adam():
    shapes = [(1000, 3), (1000, 1, 3), (1000, 15, 3), (1000, 1), (1000, 3), (1000, 4)]
    ps, gs = make(shapes, 0)
    want_p, want_m, want_v = oracle_run(ps, gs, LRS, EPS)
    params = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in ps]
    opts = [torch.optim.Adam([p], lr=lr, eps=EPS) for p, lr in zip(params, LRS)]
    for g in gs:
        for p, gi, o in zip(params, g, opts):
            p.grad = torch.from_numpy(gi.copy())
            o.step()
    for p, o, wp, wm, wv in zip(params, opts, want_p, want_m, want_v):
        st = o.state[p]
        assert close(wp, p.detach().numpy()) and close(wm, st["exp_avg"].numpy()) and close(wv, st["exp_avg_sq"].numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("shapes", [
    [(20_000, 3), (20_000, 1, 3), (20_000, 15, 3), (20_000, 1), (20_000, 3), (20_000, 4)],
    [(7,), (1,), (1023, 3), (4099,), (5, 5, 5), (2,), (3, 3), (64,), (129,), (31, 7)],   # > 8 tensors, ragged tails
])
def test_fused_adam_matches_oracle_and_torch(shapes):
    from gs_fused import FusedAdam

    lrs = (LRS * 2)[:len(shapes)]
    ps, gs = make(shapes, 1)
    want_p, want_m, want_v = oracle_run(ps, gs, lrs, EPS)
    mine = [torch.nn.Parameter(torch.from_numpy(p.copy()).cuda()) for p in ps]
    ref = [torch.nn.Parameter(torch.from_numpy(p.copy()).cuda()) for p in ps]
    opt = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(mine, lrs)], eps=EPS)
    topt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], eps=EPS)
    for g in gs:
        for p, q, gi in zip(mine, ref, g):
            p.grad = torch.from_numpy(gi.copy()).cuda()
            q.grad = p.grad.clone()
        opt.step()
        topt.step()
    for p, q, wp, wm, wv in zip(mine, ref, want_p, want_m, want_v):
        st = opt.state[p]
        assert st["step"] == len(gs)
        assert close(p.detach().cpu().numpy(), wp) and close(st["exp_avg"].cpu().numpy(), wm)
        assert close(st["exp_avg_sq"].cpu().numpy(), wv)
        assert close(p.detach().cpu().numpy(), q.detach().cpu().numpy())


@pytest.mark.gpu
def test_fused_adam_contract():
    from gs_fused import FusedAdam

    base = torch.randn(1025, device="cuda")
    p = torch.nn.Parameter(base[1:])  # contiguous but only 4-byte aligned: scalar path
    frozen = torch.nn.Parameter(torch.randn(8, device="cuda"))  # no grad: untouched
    q = torch.nn.Parameter(p.detach().clone())
    opt = FusedAdam([p, frozen], lr=0.01)
    topt = torch.optim.Adam([q], lr=0.01)
    before = frozen.detach().clone()
    for _ in range(3):
        g = torch.randn_like(p)
        p.grad, q.grad = g, g.clone()
        loss = opt.step(lambda: torch.tensor(1.5))
        topt.step()
        assert float(loss) == 1.5
    assert torch.allclose(p, q, rtol=2e-6, atol=1e-7) and torch.equal(frozen, before)
    # the state is interchangeable with torch.optim.Adam's
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    with pytest.raises(RuntimeError):  # no CPU fallback
        c = torch.nn.Parameter(torch.zeros(4))
        c.grad = torch.ones(4)
        FusedAdam([c]).step()
    with pytest.raises(ValueError):
        FusedAdam([p], betas=(1.0, 0.999))


@pytest.mark.gpu
def test_fused_adam_rematerialises_nonconforming_state():
    """State restored from a torch.optim.Adam checkpoint (or sliced by refinement) may be
    fp64 / non-contiguous: FusedAdam converts it instead of reading wrong memory."""
    import torch

    from gs_fused import FusedAdam

    torch.manual_seed(0)
    p = torch.randn(1000, 3, device="cuda", requires_grad=True)
    q = p.detach().clone().requires_grad_(True)
    ref = torch.optim.Adam([q], lr=1e-2, eps=1e-15)
    opt = FusedAdam([p], lr=1e-2, eps=1e-15)
    for it in range(3):
        g = torch.randn_like(p)
        p.grad, q.grad = g.clone(), g.clone()
        opt.step()
        ref.step()
        if it == 0:  # degrade the state: fp64 and a non-contiguous view
            st = opt.state[p]
            st["exp_avg"] = st["exp_avg"].double()
            wide = torch.zeros(1000, 6, device="cuda")
            wide[:, ::2] = st["exp_avg_sq"]
            st["exp_avg_sq"] = wide[:, ::2]
            assert not st["exp_avg_sq"].is_contiguous()
    assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)
    st = opt.state[p]
    assert st["exp_avg"].dtype == torch.float32 and st["exp_avg_sq"].is_contiguous()
    st["exp_avg"] = st["exp_avg"][:10]
    p.grad = torch.randn_like(p)
    with pytest.raises(RuntimeError):
        opt.step()


@pytest.mark.gpu
def test_fused_ops_on_random_shapes():
    """tools/exp/fuzz_fused.py: FusedAdam against torch.optim.Adam on random tensor lists (1-14 tensors, ragged
    shapes, betas incl. 0, missing and sparse-valued gradients, 1-5 steps) and the L1+SSIM head against the CPU
    oracle on random image sizes (from 11 x 11) and lambdas."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_fused.py"), "40", "61"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") == 40
