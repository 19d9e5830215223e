"""L1+SSIM head alone at 1080p (for rocprofv3): python tools/exp/loss_bench.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import torch
from gs_fused import L1SSIMLoss
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device="cuda").manual_seed(0)
gt = torch.rand(1080, 1920, 3, device="cuda", generator=g)
pred = (gt + 0.1 * torch.randn(1080, 1920, 3, device="cuda", generator=g)).clamp(0, 1).requires_grad_(True)
fn = L1SSIMLoss(0.2)
for _ in range(3):
    fn(pred, gt).backward()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(it):
    pred.grad = None
    fn(pred, gt).backward()
b.record()
torch.cuda.synchronize()
print("us per fwd+bwd", a.elapsed_time(b) / it * 1e3)
