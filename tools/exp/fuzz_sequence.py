"""Fuzz of the list cache / speculative sizing / count-free flow state machine of `rasterize_gaussians`: a random
sequence of views over scenes of different sizes, resolutions and opacities (the number of Gaussians and list
entries jumps up and down, so guesses overflow and are rebuilt) must give, call by call, exactly the images
and (up to atomic summation order) the gradients of the same calls made without speculation and without cache.
python tools/exp/fuzz_sequence.py [calls] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch
from harness import scene as S
from harness.pipeline import CameraTensors
from rasterizer import project_gaussians, rasterize_gaussians
from rasterizer import rasterize as R

DEV = torch.device("cuda:0")


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    specs = [(3000, 160, 96, 0.01, 0.1), (60_000, 640, 360, 0.01, 0.08), (250_000, 640, 360, 0.02, 0.1),
             (20_000, 2560, 1600, 0.01, 0.08), (400_000, 1280, 720, 0.002, 0.02)]
    scenes = []
    for n, W, H, lo, hi in specs:
        cam = S.make_camera(W, H)
        sc = S.make_scene(n, cam, sh_degree=0, seed=n, scale_lo=lo, scale_hi=hi)
        scenes.append((cam, {k: torch.from_numpy(v).to(DEV) for k, v in sc.items()}))
    bad = 0
    leaves = {}  # persistent opacity parameters, as a model has them: (scene, n, factor, kind) -> leaf

    def opacity_for(idx, n, ofac, mode, sc):
        """How the caller forms the opacities.  "clone": a fresh tensor every call (no provenance); "sigmoid": the
        models' `torch.sigmoid(self.opacities)` of a persistent leaf -- what lets `project_gaussians` build the lists
        ahead of time; "sigmoid_touched": the same, but the leaf is written in place (same values, new version) between
        projection and rasterisation: the lists built ahead must be dropped; "leaf": a persistent leaf handed in as is.
        -> (tensor to differentiate, opacity, touch())"""
        if mode == "clone":
            o = (sc["opacities"][:n] * ofac).clone().requires_grad_(True)
            return o, o, None
        key = (idx, n, ofac, "leaf" if mode == "leaf" else "logit")
        if key not in leaves:
            o = (sc["opacities"][:n] * ofac).clamp(1e-4, 1 - 1e-4)
            leaves[key] = (o.clone() if mode == "leaf" else torch.log(o / (1 - o))).requires_grad_(True)
        leaf = leaves[key]
        if mode == "leaf":
            return leaf, leaf, None

        def touch():
            with torch.no_grad():
                leaf.mul_(1.0)

        return leaf, None, (touch if mode == "sigmoid_touched" else None)

    def run(idx, frac, ofac, yaw, want_alpha, second, speculate, omode="clone"):
        cam0, sc = scenes[idx]
        cam = S.make_camera(cam0.width, cam0.height, yaw=yaw)
        ct = CameraTensors.from_numpy(cam, DEV)
        n = max(1, int(sc["means3d"].shape[0] * frac))
        os.environ["GSR_NO_SPECULATION"] = "0" if speculate else "1"
        if not speculate:
            R._bin_cache["key"] = None
        means = sc["means3d"][:n].clone().requires_grad_(True)
        wrt, opac, touch = opacity_for(idx, n, ofac, omode, sc)
        xys, depths, radii, conics, comp, tiles, _ = project_gaussians(
            means, sc["scales"][:n], 1.0, sc["quats"][:n], ct.viewmat[:3], ct.projmat, cam.fx, cam.fy, cam.cx, cam.cy,
            cam.height, cam.width, 16)
        if touch is not None:
            touch()
        if opac is None:
            opac = torch.sigmoid(wrt)  # formed AFTER the projection, as the models do (vanilla_gs.py:829)
        g = torch.Generator(device="cpu").manual_seed(idx * 7 + 1)
        colors = torch.rand(n, 3, generator=g).to(DEV).requires_grad_(True)
        out = rasterize_gaussians(xys, depths, radii, conics, tiles, colors, opac, cam.height, cam.width, 16,
                                  return_alpha=want_alpha)
        rgb, alpha = out if want_alpha else (out, None)
        v = torch.rand(cam.height, cam.width, 3, generator=g).to(DEV)
        loss = (rgb * v).sum() + (alpha.sum() * 0.5 if alpha is not None else 0.0)
        # the models' second call of a view (depth pass, vanilla_gs.py:840): same geometry, depths as
        # colours, and an opacity tensor that is equal in value (`second` 1: a clone -- unknown provenance,
        # compared on the device), or scaled (2: the lists must be rebuilt); 0: no second call
        depth_img = None
        if second:
            if not speculate:
                R._bin_cache["key"] = None
            opac2 = opac.detach().clone() if second == 1 else opac.detach() * 0.5
            depth_img = rasterize_gaussians(xys, depths, radii, conics, tiles, depths[:, None].repeat(1, 3), opac2,
                                            cam.height, cam.width, 16, background=torch.zeros(3, device=DEV))
            loss = loss + depth_img.sum() * 0.1
            depth_img = depth_img.detach()
        grads = torch.autograd.grad(loss, (means, wrt, colors))
        return rgb.detach(), None if alpha is None else alpha.detach(), grads, depth_img

    for k in range(calls):
        idx = int(rng.integers(len(scenes)))
        frac = float(rng.choice([1.0, 1.0, 0.5, 0.1, 0.02]))
        ofac = float(rng.choice([1.0, 1.0, 0.2, 0.01]))
        yaw = float(rng.choice([0.0, 0.0, 0.1, -0.2]))
        want_alpha = bool(rng.integers(2))
        second = int(rng.integers(3))
        omode = str(rng.choice(["clone", "sigmoid", "sigmoid", "sigmoid_touched", "leaf"]))
        if rng.integers(4) == 0 and k > 0:
            idx, frac, ofac = last  # the same scene again: the recipe AND the count hint of the previous call fit
        last = (idx, frac, ofac)
        a = run(idx, frac, ofac, yaw, want_alpha, second, True, omode)
        b = run(idx, frac, ofac, yaw, want_alpha, second, False, omode)
        ok = torch.equal(a[0], b[0]) and (a[1] is None or torch.equal(a[1], b[1]))
        ok = ok and (a[3] is None or torch.equal(a[3], b[3]))
        for ga, gb in zip(a[2], b[2]):
            ok = ok and float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-12
        spec = specs[idx]
        print(f"call {k}: scene {idx} ({int(spec[0] * frac)} Gaussians, {spec[1]}x{spec[2]}) opacity x{ofac} yaw {yaw} "
              f"alpha={want_alpha} second={second} opacity={omode}: {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    os.environ["GSR_NO_SPECULATION"] = "0"
    print("lists:", dict(R.counters))
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
