"""Fuzz of the caller-side fused ops: FusedAdam against torch.optim.Adam (random tensor counts and shapes,
learning rates, betas, eps, missing gradients, steps) and the L1+SSIM loss head against the CPU oracle (random
image sizes from 11 x 11, lambda, clamp).  python tools/exp/fuzz_fused.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch
from gs_fused import FusedAdam, L1SSIMLoss
from oracle import oracle as O


def close(a, b, rel=4e-6, floor=1e-30):
    return float((a - b).abs().max()) <= rel * max(floor, float(b.abs().max()))


def adam_case(rng):
    nt = int(rng.integers(1, 15))
    shapes = []
    for _ in range(nt):
        d = int(rng.integers(1, 4))
        shape = tuple(int(x) for x in rng.choice([1, 2, 3, 5, 16, 63, 257, 4099], d))
        while int(np.prod(shape)) > 2_000_000:  # keep a case in the megabytes
            shape = shape[:-1]
        shapes.append(shape)
    betas = (float(rng.choice([0.9, 0.5, 0.0])), float(rng.choice([0.999, 0.9])))
    eps = float(rng.choice([1e-15, 1e-8]))
    lrs = [float(10.0 ** rng.uniform(-5, -1)) for _ in shapes]
    g = torch.Generator(device="cpu").manual_seed(int(rng.integers(1 << 30)))
    mine = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(mine, lrs)], betas=betas, eps=eps)
    topt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], betas=betas, eps=eps)
    for step in range(int(rng.integers(1, 6))):
        for p, q in zip(mine, ref):
            if rng.random() < 0.15:
                p.grad = q.grad = None  # a parameter without gradient this step: untouched, step not advanced
                continue
            gr = (torch.randn(p.shape, generator=g) * float(10.0 ** rng.uniform(-6, 0))).cuda()
            if rng.random() < 0.2:
                gr[gr.abs() < gr.abs().median()] = 0.0
            p.grad, q.grad = gr, gr.clone()
        opt.step()
        topt.step()
    for p, q, lr in zip(mine, ref, lrs):
        # (an Adam step moves a parameter by ~lr: a 1-ulp difference in the step is 1e-7 lr, which a parameter that
        #  the step happens to bring near zero would otherwise be held against -- seed 3003, case 60)
        assert close(p.detach(), q.detach(), floor=lr), "parameter"
        a, b = opt.state.get(p, {}), topt.state.get(q, {})
        assert ("exp_avg" in a) == ("exp_avg" in b)
        if "exp_avg" in a:
            assert close(a["exp_avg"], b["exp_avg"]) and close(a["exp_avg_sq"], b["exp_avg_sq"]), "moments"
            assert int(a["step"]) == int(b["step"])
    return f"adam {nt} tensors betas={betas} eps={eps}"


def loss_case(rng):
    H, W = int(rng.integers(11, 320)), int(rng.integers(11, 420))
    lam = float(rng.choice([0.0, 0.2, 0.8, 1.0]))
    gt = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    pred = (gt + float(rng.choice([0.01, 0.1, 0.5])) * rng.standard_normal((H, W, 3))).astype(np.float32)
    pred = np.clip(pred, 0, 1).astype(np.float32)
    p = torch.from_numpy(pred).cuda().requires_grad_(True)
    loss = L1SSIMLoss(lam)(p, torch.from_numpy(gt).cuda())
    loss.backward()
    ref_loss, _, _, v = O.l1_ssim_loss(pred, gt, lam)
    assert abs(float(loss) - ref_loss) < 3e-6, f"loss {float(loss)} vs {ref_loss}"
    got = p.grad.cpu().numpy()
    stable = np.abs(pred.astype(np.float64) - gt) > 1e-6  # the L1 sign term flips where pred == gt
    assert np.abs(got - v)[stable].max() < 1e-4 * np.abs(v).max() + 1e-9, "gradient"
    return f"loss {H}x{W} lambda={lam}"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for k in range(cases):
        fn = adam_case if k % 2 == 0 else loss_case
        try:
            print(f"case {k}: {fn(rng)} ok", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"case {k}: {fn.__name__} MISMATCH {str(e)[:200]}", flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
