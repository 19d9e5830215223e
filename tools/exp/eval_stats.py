import os, sys, ctypes, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
import harness.scene as S
from harness.pipeline import render_view, CameraTensors
from rasterizer.cuda._backend import lib
dev = torch.device("cuda", 0)
for name, kw in (("default", dict(scale_lo=0.0025, scale_hi=0.025)), ("dense", dict(scale_lo=0.005, scale_hi=0.05)), ("longtail", dict(scale_lo=0.0025, scale_hi=0.025, longtail=True))):
    W, H, N = 1920, 1080, 1_000_000
    cam = S.make_camera(W, H); sc = S.make_scene(N, cam, sh_degree=3, seed=42, **kw)
    t = lambda a: torch.from_numpy(a).to(dev)
    params = {k: t(v).requires_grad_(True) for k, v in sc.items()}
    camt = CameraTensors.from_numpy(cam, dev); bg = t(np.array(S.BACKGROUND, np.float32))
    v_img, v_alpha = (t(a) for a in S.make_cotangents(cam))
    def step():
        out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"], params["sh_coeffs"], camt, bg, 3, clamp_rgb=False)
        torch.autograd.backward([out["rgb"], out["alpha"]], [v_img, v_alpha[..., None]])
    step(); step()
    # (needs the temporary counters of DESIGN 4.2 "How much of the backward's work is useful" compiled into raster_bwd.hip: staged + 1 .. + 3)
    c = torch.zeros(8, dtype=torch.int64, device=dev); torch.cuda.synchronize()
    lib().gsr_debug_count_staged(ctypes.c_void_p(c.data_ptr())); step(); torch.cuda.synchronize(); lib().gsr_debug_count_staged(None)
    fwd, bwd, ev, useful, px = c[:5].tolist()
    print(f"{name}: staged fwd {fwd} bwd {bwd}; bwd quadrant evaluations {ev} ({ev/max(bwd,1):.2f} per entry), with a contributing pixel {useful} ({useful/max(ev,1):.1%}), contributing pixel-splat pairs {px} ({px/max(ev*64,1):.1%} of evaluated lanes)")
