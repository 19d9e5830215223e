"""cProfile of the host side of one small fwd+bwd step (where do the ~0.7 ms of config 1 go?).
python tools/exp/host_prof.py [gaussians] [W] [H] [deg]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch
from harness import scene as S
from harness.pipeline import CameraTensors, render_view

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
deg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
cam = S.make_camera(W, H)
sc = S.make_scene(N, cam, sh_degree=deg, seed=42, scale_lo=0.005, scale_hi=0.05)
params = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
camt = CameraTensors.from_numpy(cam, dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
v_img, v_alpha = [torch.from_numpy(a).to(dev) for a in S.make_cotangents(cam)]


def step():
    for p in params.values():
        p.grad = None
    out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"], params["sh_coeffs"],
                      camt, bg, deg, clamp_rgb=False)
    torch.autograd.backward([out["rgb"], out["alpha"]], [v_img, v_alpha[..., None]])


for _ in range(20):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t) / 200 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
