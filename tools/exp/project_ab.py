"""Kernel time of the projection (forward, and backward) on bench.py's default scene, by HIP events.
   python tools/exp/project_ab.py [reps]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import torch
from rasterizer import cuda as C

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
from harness.scene import make_scene, make_camera
cam = make_camera(1920, 1080)
sc = {k: torch.from_numpy(v).to(dev) for k, v in make_scene(1_000_000, cam, scale_lo=0.0025, scale_hi=0.025).items()}
T = lambda x: torch.from_numpy(x).to(dev).contiguous()
args = (1_000_000, sc["means3d"], sc["scales"], 1.0, sc["quats"], T(cam.viewmat), T(cam.projmat), cam.fx, cam.fy,
        cam.cx, cam.cy, cam.height, cam.width, 16, 0.01)

def timed(f):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

out = C.project_gaussians_forward(*args)
print("project forward  %.2f us per call (incl. launch + carve)" % timed(lambda: C.project_gaussians_forward(*args)))
import hashlib
h = hashlib.sha256()
for t in out: h.update(t.detach().cpu().numpy().tobytes())
print("outputs sha256", h.hexdigest()[:16], "visible", int((out[3] > 0).sum()))
