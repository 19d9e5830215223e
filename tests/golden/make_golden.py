#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own
pure-PyTorch implementation (`rasterizer/_torch_impl.py`).

Runs ONLY in the build container (it imports /root/reference, which does not
exist on the GPU box).  The committed `.npz` files are data: inputs and the
reference's outputs / autograd gradients.  Nothing of the reference's source
is stored.

    python tests/golden/make_golden.py            # regenerate everything

Quirks of the reference implementation that shape the fixtures:
  * `_torch_impl.map_gaussian_to_intersects` stops at the first culled
    Gaussian -> every scene is permuted so that visible Gaussians come first;
  * `_torch_impl.rasterize_forward` needs tile 0 to be non-empty -> every
    scene has a Gaussian pinned on the top-left corner;
  * `final_idx` has a different meaning there -> not stored;
  * the oracle wants a 4x4 view matrix; the CUDA op reads the top 3x4 only;
  * opacities <= 0.95 so the forward (0.999) / backward (0.99) alpha clamps of
    the CUDA kernels do not engage (autograd has neither asymmetry);
  * visible Gaussians stay inside the 1.3x fov guard band (the CUDA backward
    ignores the derivative of that clamp);
  * all depths are distinct (tie order of torch.sort is unspecified);
  * I >= 2 (the Python get_tile_bin_edges mishandles a single intersection).
"""
import math
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gs_toolkit/gs_components"


def _import_reference():
    shim = tempfile.mkdtemp(prefix="jaxtyping_shim_")
    with open(os.path.join(shim, "jaxtyping.py"), "w") as f:
        f.write(
            "class _T:\n"
            "    def __class_getitem__(cls, item):\n"
            "        return cls\n"
            "class Float(_T): pass\n"
            "class Int(_T): pass\n"
        )
    sys.path.insert(0, shim)
    sys.path.insert(0, REF)
    import rasterizer._torch_impl as ti  # noqa: E402

    assert ti.__file__.startswith(REF), ti.__file__
    return ti


def projection_matrix(znear, zfar, fovx, fovy):
    # gs_toolkit/utils/comms.py:103-123 (OpenGL-style, w_clip = z_view)
    t = znear * math.tan(0.5 * fovy)
    b = -t
    r = znear * math.tan(0.5 * fovx)
    l = -r
    n, f = znear, zfar
    return torch.tensor(
        [
            [2 * n / (r - l), 0.0, (r + l) / (r - l), 0.0],
            [0.0, 2 * n / (t - b), (t + b) / (t - b), 0.0],
            [0.0, 0.0, (f + n) / (f - n), -1.0 * f * n / (f - n)],
            [0.0, 0.0, 1.0, 0.0],
        ],
        dtype=torch.float32,
    )


def camera(W, H, fov_x_deg=60.0, rot=None, trans=None):
    fovx = math.radians(fov_x_deg)
    fx = W / (2 * math.tan(fovx / 2))
    fy = fx
    fovy = 2 * math.atan(H / (2 * fy))
    cx, cy = W / 2.0, H / 2.0
    viewmat = torch.eye(4)
    if rot is not None:
        ax, ay, az = rot
        Rx = torch.tensor([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
        Ry = torch.tensor([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
        Rz = torch.tensor([[math.cos(az), -math.sin(az), 0], [math.sin(az), math.cos(az), 0], [0, 0, 1]])
        viewmat[:3, :3] = (Rz @ Ry @ Rx).float()
    if trans is not None:
        viewmat[:3, 3] = torch.tensor(trans, dtype=torch.float32)
    projmat = projection_matrix(0.001, 1000.0, fovx, fovy) @ viewmat
    return dict(W=W, H=H, fx=fx, fy=fy, cx=cx, cy=cy, viewmat=viewmat, projmat=projmat,
                tanx=math.tan(fovx / 2), tany=math.tan(fovy / 2))


def cam_to_world(cam, p_cam):
    V = cam["viewmat"]
    R, t = V[:3, :3], V[:3, 3]
    return (p_cam - t) @ R  # R^T (p - t) for row vectors


def random_scene(n, cam, zlo, zhi, slo, shi, spread=0.9, gen=None):
    z = torch.linspace(zlo, zhi, n)[torch.randperm(n, generator=gen)]
    z = z + 1e-3 * torch.rand(n, generator=gen)  # distinct depths
    x = (torch.rand(n, generator=gen) * 2 - 1) * spread * cam["tanx"] * z
    y = (torch.rand(n, generator=gen) * 2 - 1) * spread * cam["tany"] * z
    p_cam = torch.stack([x, y, z], -1)
    means = cam_to_world(cam, p_cam)
    scales = torch.exp(torch.rand(n, 3, generator=gen) * (math.log(shi) - math.log(slo)) + math.log(slo))
    quats = torch.nn.functional.normalize(torch.randn(n, 4, generator=gen), dim=-1)
    opac = torch.rand(n, 1, generator=gen) * 0.85 + 0.1
    colors = torch.rand(n, 3, generator=gen)
    return means, scales, quats, opac, colors


def pin_corner(cam, means, scales, quats, opac, z=1.0, s=0.05):
    """Gaussian 0 projects to pixel ~(1,1) so tile 0 is never empty."""
    W, H = cam["W"], cam["H"]
    u, v = 1.0, 1.0
    xc = (u + 0.5 - cam["cx"]) / cam["fx"] * z
    yc = (v + 0.5 - cam["cy"]) / cam["fy"] * z
    means[0] = cam_to_world(cam, torch.tensor([[xc, yc, z]]))[0]
    scales[0] = torch.tensor([s, s, s])
    quats[0] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    opac[0] = 0.5


def run_scene(ti, name, cam, means, scales, quats, opac, colors, background, bw=16, glob_scale=1.0):
    t0 = time.time()
    W, H = cam["W"], cam["H"]
    intr = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    viewmat, projmat = cam["viewmat"], cam["projmat"]

    # pass 1: find the visible set, permute visible-first (see module docstring)
    with torch.no_grad():
        out = ti.project_gaussians_forward(means, scales, glob_scale, quats, viewmat, projmat, intr, (W, H), bw)
        mask = out[-1]
    order = torch.cat([torch.nonzero(mask)[:, 0], torch.nonzero(~mask)[:, 0]])
    means, scales, quats, opac, colors = (t[order].clone() for t in (means, scales, quats, opac, colors))

    means.requires_grad_(True)
    scales.requires_grad_(True)
    quats.requires_grad_(True)
    opac.requires_grad_(True)
    colors.requires_grad_(True)
    (cov3d, cov2d, xys, depths, radii, conic, comp, num_tiles_hit, mask) = ti.project_gaussians_forward(
        means, scales, glob_scale, quats, viewmat, projmat, intr, (W, H), bw
    )
    xys.retain_grad()
    conic.retain_grad()
    n = means.shape[0]
    tile_bounds = ((W + bw - 1) // bw, (H + bw - 1) // bw, 1)
    cum = torch.cumsum(num_tiles_hit, dim=0, dtype=torch.int32)
    I = int(cum[-1].item())
    with torch.no_grad():
        isect, gids = ti.map_gaussian_to_intersects(n, xys, depths, radii, cum, tile_bounds, bw)
        ks, idx = torch.sort(isect)
        vs = torch.gather(gids, 0, idx)
        assert len(torch.unique(ks)) == len(ks), "duplicate keys -> tie order unspecified"
        bins = ti.get_tile_bin_edges(I, ks, tile_bounds)
    assert bins[0, 1] > bins[0, 0], "tile 0 must be non-empty for the reference loop"
    out_img, final_Ts, _ = ti.rasterize_forward(
        tile_bounds, (bw, bw, 1), (W, H, 1), vs, bins, xys, conic, colors, opac[:, 0], background
    )
    gen = torch.Generator().manual_seed(1234)
    w = torch.rand(H, W, 3, generator=gen) * 2 - 1
    u = torch.rand(H, W, generator=gen) * 2 - 1
    alpha = 1 - final_Ts
    loss = (out_img * w).sum() + (alpha * u).sum()
    loss.backward()
    for t in (means, scales, quats, opac, colors, xys, conic):
        assert torch.isfinite(t.grad).all()

    def a(t, dt=None):
        t = t.detach().cpu().numpy()
        return t.astype(dt) if dt is not None else t

    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"),
        # inputs
        means3d=a(means), scales=a(scales), quats=a(quats), opacities=a(opac), colors=a(colors),
        background=a(background), viewmat=a(viewmat), projmat=a(projmat),
        intrinsics=np.array(intr, np.float64), img_size=np.array([W, H], np.int32),
        block_width=np.int32(bw), glob_scale=np.float32(glob_scale),
        # project outputs
        cov3d=a(cov3d), cov2d=a(cov2d), xys=a(xys), depths=a(depths), radii=a(radii, np.int32),
        conics=a(conic), compensation=a(comp), num_tiles_hit=a(num_tiles_hit, np.int32),
        mask=a(mask),
        # binning
        isect_ids=a(isect), gaussian_ids=a(gids, np.int32), isect_ids_sorted=a(ks),
        gaussian_ids_sorted=a(vs, np.int32), tile_bins=a(bins, np.int32),
        # composite
        out_img=a(out_img), final_Ts=a(final_Ts),
        # cotangents + autograd gradients
        v_out_img=a(w), v_out_alpha=a(u),
        g_means3d=a(means.grad), g_scales=a(scales.grad), g_quats=a(quats.grad),
        g_opacities=a(opac.grad), g_colors=a(colors.grad), g_xys=a(xys.grad), g_conics=a(conic.grad),
    )
    print(f"{name}: N={n} visible={int(mask.sum())} I={I} {W}x{H}  {time.time() - t0:.1f}s")


def make_sh(ti):
    gen = torch.Generator().manual_seed(7)
    n = 64
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    out = {"viewdirs": dirs.numpy()}
    for deg in range(5):
        K = (deg + 1) ** 2
        coeffs = torch.randn(n, K, 3, generator=gen)
        coeffs.requires_grad_(True)
        colors = ti.compute_sh_color(dirs, coeffs)
        v = torch.rand(n, 3, generator=gen) * 2 - 1
        (colors * v).sum().backward()
        out[f"coeffs{deg}"] = coeffs.detach().numpy()
        out[f"colors{deg}"] = colors.detach().numpy()
        out[f"v_colors{deg}"] = v.numpy()
        out[f"g_coeffs{deg}"] = coeffs.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "sh.npz"), **out)
    print("sh: degrees 0..4, n=64")


def main():
    ti = _import_reference()
    torch.manual_seed(0)
    make_sh(ti)

    # g0: two Gaussians near the optical axis, 16x16 (one tile), identity view.
    # (one Gaussian alone gives I=1, for which the reference's Python
    # get_tile_bin_edges never closes the bin -- _torch_impl.py:386-392)
    cam = camera(16, 16)
    means = torch.tensor([[0.0, 0.0, 2.0], [0.1, -0.05, 3.0]])
    scales = torch.tensor([[0.15, 0.1, 0.12], [0.3, 0.2, 0.25]])
    quats = torch.nn.functional.normalize(torch.tensor([[0.9, 0.1, -0.2, 0.3], [0.5, -0.5, 0.3, 0.1]]), dim=-1)
    opac = torch.tensor([[0.8], [0.6]])
    colors = torch.tensor([[0.9, 0.4, 0.2], [0.1, 0.7, 0.5]])
    run_scene(ti, "g0", cam, means, scales, quats, opac, colors, torch.tensor([0.1, 0.2, 0.3]))

    # g1: 8 Gaussians incl. behind-camera, off-screen, huge; 32x32 and 40x24; rotated camera
    for tag, (W, H) in (("g1a", (32, 32)), ("g1b", (40, 24))):
        gen = torch.Generator().manual_seed(11)
        cam = camera(W, H, rot=(0.1, -0.15, 0.05), trans=(0.2, -0.1, 0.3))
        means, scales, quats, opac, colors = random_scene(8, cam, 1.5, 4.0, 0.03, 0.2, gen=gen)
        pin_corner(cam, means, scales, quats, opac)
        means[1] = cam_to_world(cam, torch.tensor([[0.1, 0.1, -1.0]]))[0]  # behind the camera
        means[2] = cam_to_world(cam, torch.tensor([[30.0, 0.0, 2.0]]))[0]  # far off-screen
        means[3] = cam_to_world(cam, torch.tensor([[0.05, -0.05, 3.0]]))[0]
        scales[3] = torch.tensor([1.5, 1.2, 0.9])  # huge
        run_scene(ti, tag, cam, means, scales, quats, opac, colors, torch.tensor([0.3, 0.1, 0.6]))

    # g2: 64 Gaussians, 48x48
    gen = torch.Generator().manual_seed(22)
    cam = camera(48, 48, rot=(-0.05, 0.1, -0.2), trans=(-0.1, 0.15, 0.2))
    means, scales, quats, opac, colors = random_scene(64, cam, 1.5, 6.0, 0.02, 0.15, gen=gen)
    pin_corner(cam, means, scales, quats, opac)
    run_scene(ti, "g2", cam, means, scales, quats, opac, colors, torch.tensor([0.149, 0.1647, 0.2157]))

    # g3: 200 Gaussians, 64x64, dense overlap -> T<=1e-4 termination is exercised
    gen = torch.Generator().manual_seed(33)
    cam = camera(64, 64)
    means, scales, quats, opac, colors = random_scene(200, cam, 1.0, 5.0, 0.05, 0.3, spread=0.8, gen=gen)
    opac = opac.clamp(min=0.6, max=0.95)
    pin_corner(cam, means, scales, quats, opac)
    run_scene(ti, "g3", cam, means, scales, quats, opac, colors, torch.tensor([1.0, 1.0, 1.0]))


if __name__ == "__main__":
    main()
