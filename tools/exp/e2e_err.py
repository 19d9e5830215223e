"""Where does the end-to-end gradient error (GPU pipeline vs oracle pipeline) come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gaussian-splatting-toolkit_amd')]
import numpy as np, torch
from harness import scene as S
from harness.pipeline import CameraTensors, render_view
from oracle import oracle as O
DEV = 'cuda:0'
cu = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(g)
npy = lambda t: t.detach().cpu().numpy()

def run(n, W, H, deg, seed, lo, hi, cam_kw={}):
    cam = S.make_camera(W, H, **cam_kw)
    sc = S.make_scene(n, cam, sh_degree=deg, seed=seed, scale_lo=lo, scale_hi=hi)
    bg = np.array(S.BACKGROUND, np.float32)
    v_img, v_alpha = S.make_cotangents(cam)
    params = {k: cu(v, True) for k, v in sc.items()}
    out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"], params["sh_coeffs"],
                      CameraTensors.from_numpy(cam, DEV), cu(bg), deg, retain_xys_grad=True, clamp_rgb=False)
    loss = (out["rgb"] * cu(v_img)).sum() + (out["alpha"][..., 0] * cu(v_alpha)).sum()
    loss.backward()
    dirs = S.viewdirs_for(sc, cam)
    sh = O.compute_sh_forward(n, deg, deg, dirs, sc["sh_coeffs"])
    rgbs = np.maximum(sh + 0.5, 0).astype(np.float32)
    r = O.render_forward(sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat, cam.fx, cam.fy,
                         cam.cx, cam.cy, H, W, 16, rgbs, sc["opacities"], bg, ambig_eps=1e-5)
    res = O.rasterize_backward(H, W, 16, r["gaussian_ids_sorted"], r["tile_bins"], r["xys"], r["conics"], rgbs,
                               sc["opacities"], bg, r["final_Ts"], r["final_idx"], v_img, v_alpha, with_abs_sums=True,
                               ambig_eps=1e-5)
    vxy, vconic, vcol, vop = res[:4]
    axy = res[4]
    amb_g = res[8]
    # Gaussians whose 3-sigma box covers a forward-ambiguous pixel
    amb_px = r["ambig"].astype(np.int64)
    ii = np.zeros((H + 1, W + 1), np.int64); ii[1:, 1:] = amb_px.cumsum(0).cumsum(1)
    x, y, rad = r["xys"][:, 0], r["xys"][:, 1], r["radii"].astype(np.float32)
    x0 = np.clip(np.floor(x - rad), 0, W).astype(int); x1 = np.clip(np.ceil(x + rad) + 1, 0, W).astype(int)
    y0 = np.clip(np.floor(y - rad), 0, H).astype(int); y1 = np.clip(np.ceil(y + rad) + 1, 0, H).astype(int)
    touched = (ii[y1, x1] - ii[y0, x1] - ii[y1, x0] + ii[y0, x0]) > 0
    stable = ~(amb_g | touched) 
    g = npy(out["xys"].grad)
    err = np.abs(g - vxy); mx = np.abs(vxy).max()
    print(f"n={n} {W}x{H}: amb px {amb_px.mean():.4f}, amb gauss {amb_g.mean():.4f}, touched {touched.mean():.4f}, stable {stable.mean():.4f}")
    print("  xys.grad  max err/max|ref| all %.2e  stable %.2e ; rel(floor 1e-3 max) all %.2e stable %.2e ; err/abs_sum stable %.2e all %.2e" % (
        err.max() / mx, err[stable].max() / mx, (err / np.maximum(np.abs(vxy), 1e-3 * mx)).max(),
        (err / np.maximum(np.abs(vxy), 1e-3 * mx))[stable].max(), (err / np.maximum(axy, 1e-30))[stable].max(), (err / np.maximum(axy, 1e-30)).max()))
    i = np.unravel_index((err * stable[:, None]).argmax(), err.shape)
    print("  worst stable", i, "ref", vxy[i], "got", g[i], "abs_sum", axy[i], "radius", r["radii"][i[0]], "opac", sc["opacities"][i[0], 0])

run(2000, 128, 96, 3, 7, 0.01, 0.1, dict(yaw=0.1, pitch=-0.05))
run(10_000, 256, 256, 0, 42, 0.005, 0.05)
run(200_000, 1920, 1080, 3, 42, 0.005, 0.05)
