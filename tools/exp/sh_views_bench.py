"""gsr_sh_backward_views next to the single-view SH backward it replaces under data parallelism: 1 M Gaussians,
degree 3, W = 1 / 2 / 4 / 8 views read out of one gathered [W, 3 N + 3] message."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
import rasterizer.cuda as C
from gs_fused import sh_backward_views

n, deg = 1_000_000, 3
rng = np.random.default_rng(0)
means = torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)).cuda()
def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = {}
d = means / means.norm(dim=-1, keepdim=True)
v = torch.randn(n, 3, device="cuda")
out["sh_backward_one_view_us"] = round(timed(lambda: C.compute_sh_backward(n, deg, deg, d, v)), 1)
for W in (1, 2, 4, 8):
    msg = torch.randn(W, 3 * n + 3, device="cuda")
    msg[:, 3 * n:] = torch.rand(W, 3, device="cuda") + 4
    for split in (True, False):
        us = timed(lambda: sh_backward_views(deg, deg, means, msg[:, 3 * n:], msg[:, :3 * n], 1.0 / W, split=split))
        out[f"views_W{W}_{'split' if split else 'joint'}_us"] = round(us, 1)
print(json.dumps(out))
