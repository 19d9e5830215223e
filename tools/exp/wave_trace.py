"""Who ran when: the wave timeline of ONE compositing forward and backward launch (gsr_debug_wave_trace).

    python tools/exp/wave_trace.py [--scene uniform|longtail|ply:<path>] [--gaussians N] [--width W --height H]

Every wave of the 16x16 compositing kernels records the constant 100-MHz clock at entry and exit, its tile, its sub-tile
mask, the tile's list length and its hardware slot.  From that: the launch's span, how many waves were resident over
time (the tail), how evenly the SIMDs were loaded, how duration relates to list length, and what a perfectly packed
schedule of the same waves would take -- the attribution VERDICT r4 asked for before acting on the trained
distribution's VALU-busy 0.52.  GSR_TUNE overrides of the tuning table apply."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from harness import scene as S  # noqa: E402
from harness.pipeline import CameraTensors, render_view  # noqa: E402


def analyse(name, rec, slots_per_simd):
    rec = rec[rec[:, 1] > 0]
    if not len(rec):
        print(name, ": no records")
        return
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    tile, allowed = (rec[:, 2] & 0xffffffff).astype(np.int64), (rec[:, 2] >> 32).astype(np.int64)
    length, hw = (rec[:, 3] & 0xffffffff).astype(np.int64), (rec[:, 3] >> 32).astype(np.int64)
    simd = ((hw >> 20) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf) * 16 + ((hw >> 4) & 3)
    begin, end = t0.min(), t1.max()
    span = (end - begin) / 100.0  # us
    dur = (t1 - t0) / 100.0
    n_simd = len(np.unique(simd))
    print(f"== {name}: {len(rec)} waves on {n_simd} SIMDs, span {span:.1f} us, wave time sum {dur.sum() / 1e3:.2f} ms "
          f"= {dur.sum() / span / max(n_simd, 1):.2f} resident waves per SIMD on average (slots: {slots_per_simd})")
    split = allowed != 15
    print(f"   whole-tile waves {int((~split).sum())}, sub-tile waves {int(split.sum())}; list length p50 {int(np.median(length))} "
          f"p90 {int(np.percentile(length, 90))} max {int(length.max())}")
    print(f"   wave duration us: p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} "
          f"max {dur.max():.1f};  us per list entry (whole-tile waves): {np.median(dur[~split] / np.maximum(length[~split], 1)):.4f}"
          + (f", (sub-tile waves): {np.median(dur[split] / np.maximum(length[split], 1)):.4f}" if split.any() else ""))
    # resident waves over time
    bins = 20
    edges = np.linspace(begin, end, bins + 1)
    occ = []
    for a, b in zip(edges[:-1], edges[1:]):
        occ.append(np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a))
    print("   resident waves per SIMD over the span (20 slices): " + " ".join(f"{o / max(n_simd, 1):.1f}" for o in occ))
    # when did the last wave START, and how long after the median end did the launch end
    print(f"   last wave starts at {100 * (t0.max() - begin) / (end - begin):.0f} % of the span; 50 % of the waves have ended by "
          f"{100 * (np.median(t1) - begin) / (end - begin):.0f} %, 90 % by {100 * (np.percentile(t1, 90) - begin) / (end - begin):.0f} %, "
          f"99 % by {100 * (np.percentile(t1, 99) - begin) / (end - begin):.0f} %")
    # per-SIMD busy time (union is not needed: sum of wave time per SIMD / slots)
    per = np.bincount(np.unique(simd, return_inverse=True)[1], weights=dur)
    print(f"   wave time per SIMD: min {per.min():.0f} p50 {np.median(per):.0f} max {per.max():.0f} us (span {span:.0f})")
    late = np.argsort(-t1)[:6]
    print("   the six waves that end last: " + "; ".join(
        f"tile {tile[i]} mask {allowed[i]} len {length[i]} start {100 * (t0[i] - begin) / (end - begin):.0f}% dur {dur[i]:.0f}us" for i in late))
    return span


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="uniform")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--ply-cam-radius", type=float, default=5.0)
    ap.add_argument("--ply-view", type=int, default=0)
    ap.add_argument("--out", default=None, help="save the raw records (npz)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    W, H, deg = args.width, args.height, 3
    if args.scene.startswith("ply:"):
        from gs_io.ply import read_gaussian_ply
        from harness.train import orbit_cameras

        raw = read_gaussian_ply(args.scene[4:])
        deg = {0: 0, 3: 1, 8: 2, 15: 3}[raw["features_rest"].shape[1]]
        q = raw["quats"] / np.linalg.norm(raw["quats"], axis=-1, keepdims=True)
        sc = {"means3d": raw["means"], "scales": np.exp(raw["scales"]).astype(np.float32), "quats": q.astype(np.float32),
              "opacities": (1.0 / (1.0 + np.exp(-raw["opacities"].astype(np.float64)))).astype(np.float32),
              "sh_coeffs": np.ascontiguousarray(np.concatenate([raw["features_dc"][:, None, :], raw["features_rest"]], 1))}
        cam = orbit_cameras(48, W, H, radius=args.ply_cam_radius)[args.ply_view]
    elif args.scene == "ball":  # the trainer's default scene (harness.train.blob_scene): a ball of translucent Gaussians
        from harness.train import blob_scene, orbit_cameras

        raw = blob_scene(args.gaussians, seed=0, sh_degree=3)
        q = raw["quats"] / np.linalg.norm(raw["quats"], axis=-1, keepdims=True)
        sc = {"means3d": raw["means"], "scales": np.exp(raw["scales"]).astype(np.float32), "quats": q.astype(np.float32),
              "opacities": (1.0 / (1.0 + np.exp(-raw["opacities"].astype(np.float64) + 0.5))).astype(np.float32),
              "sh_coeffs": np.ascontiguousarray(np.concatenate([raw["features_dc"][:, None, :], raw["features_rest"]], 1))}
        cam = orbit_cameras(16, W, H, radius=6.0)[args.ply_view]
    else:
        cam = S.make_camera(W, H)
        sc = S.make_scene(args.gaussians, cam, sh_degree=deg, seed=42, scale_lo=0.0025, scale_hi=0.025,
                          longtail=args.scene == "longtail")
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    params = {k: t(v).requires_grad_(True) for k, v in sc.items()}
    camt = CameraTensors.from_numpy(cam, dev)
    v_img_np, v_alpha_np = S.make_cotangents(cam)
    bg, v_img, v_alpha = t(np.array(S.BACKGROUND, np.float32)), t(v_img_np), t(v_alpha_np)

    def step():
        for p in params.values():
            p.grad = None
        out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"], params["sh_coeffs"],
                          camt, bg, deg, clamp_rgb=False)
        torch.autograd.backward([out["rgb"], out["alpha"]], [v_img, v_alpha[..., None]])

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from rasterizer.cuda._backend import lib

    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    cap = 64 * (((tiles + 7) // 8) * 8 * 4 + 64)  # (room for a segmented grid)
    buf = torch.zeros((2 * cap, 4), dtype=torch.int64, device=dev)
    if lib().gsr_debug_wave_trace(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint(cap)) != 0:
        raise SystemExit(lib().gsr_last_error().decode())
    step()
    torch.cuda.synchronize()
    lib().gsr_debug_wave_trace(None, ctypes.c_uint(0))
    rec = buf.cpu().numpy().view(np.uint64)
    print(f"scene {args.scene}, {sc['means3d'].shape[0]} Gaussians, {W}x{H} ({tiles} tiles); GSR_TUNE={os.environ.get('GSR_TUNE', '{}')}")
    analyse("forward ", rec[:cap], 8)
    analyse("backward", rec[cap:], 4)
    if args.out:
        np.savez_compressed(args.out, fwd=rec[:cap][rec[:cap, 1] > 0], bwd=rec[cap:][rec[cap:, 1] > 0])


if __name__ == "__main__":
    main()
