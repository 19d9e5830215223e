"""cProfile of the host side of forward-only frames (viewer path): python tools/exp/host_prof_fwd.py [--depth]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import torch
from harness import scene as S
from harness.pipeline import CameraTensors, render_view

depth = "--depth" in sys.argv
dev = torch.device("cuda:0")
cam = S.make_camera(1920, 1080)
sc = S.make_scene(1_000_000, cam, sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
p = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
camt = CameraTensors.from_numpy(cam, dev)
bg = torch.tensor(S.BACKGROUND, device=dev)


def frame():
    with torch.no_grad():
        return render_view(p["means3d"], p["scales"], p["quats"], p["opacities"], p["sh_coeffs"], camt, bg, 3,
                           render_depth=depth, fused_depth=depth)


for _ in range(20):
    frame()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    frame()
t_host = (time.perf_counter() - t) / 200 * 1e3
torch.cuda.synchronize()
print("ms/frame host-side issue", t_host, " total", (time.perf_counter() - t) / 200 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    frame()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
