#!/usr/bin/env python3
"""Times one optimiser step over the 59 floats/Gaussian of a 1 M-Gaussian SH3 model:
gs_fused.FusedAdam vs torch.optim.Adam(fused=True) vs six torch.optim.Adam."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
from gs_fused import FusedAdam  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
shapes = [(N, 3), (N, 1, 3), (N, 15, 3), (N, 1), (N, 3), (N, 4)]
lrs = [1.6e-4, 0.0025, 0.0025 / 20, 0.05, 0.005, 0.001]


def run(make_opt, name):
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-3
    opts = make_opt(ps)
    for _ in range(3):
        for o in opts:
            o.step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        for o in opts:
            o.step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    nbytes = sum(p.numel() for p in ps) * 28
    print(f"{name}: {ms * 1e3:.0f} us/step  ({nbytes / ms / 1e6:.0f} GB/s of the 28 B/element)")


groups = lambda ps: [{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)]
run(lambda ps: [FusedAdam(groups(ps), eps=1e-15)], "gs_fused.FusedAdam (1 launch)")
run(lambda ps: [torch.optim.Adam(groups(ps), eps=1e-15, fused=True)], "torch Adam fused=True")
run(lambda ps: [torch.optim.Adam([p], lr=lr, eps=1e-15) for p, lr in zip(ps, lrs)], "6 x torch.optim.Adam")
