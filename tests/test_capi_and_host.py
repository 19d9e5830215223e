"""CPU: the C-ABI library loads and exports every symbol include/gsraster.h
declares; host-side logic of the drop-in package that needs no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "gsraster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for s in ("gsr_project_forward", "gsr_project_backward", "gsr_sh_forward", "gsr_sh_backward",
              "gsr_cumsum_tiles", "gsr_map_intersects", "gsr_sort_intersects", "gsr_tile_bin_edges",
              "gsr_rasterize_forward", "gsr_rasterize_backward", "gsr_rasterize_forward_nd",
              "gsr_rasterize_backward_nd", "gsr_cov2d_bounds", "gsr_last_error", "gsr_version"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from rasterizer.cuda import _backend

    assert os.path.exists(_backend.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(_backend.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in gsraster.h but not exported"
    assert sorted(_backend.SYMBOLS) == header_symbols()
    lib.gsr_version.restype = ctypes.c_int
    assert lib.gsr_version() == 200
    lib.gsr_last_error.restype = ctypes.c_char_p
    assert lib.gsr_last_error() == b""


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host before anything touches a device."""
    from rasterizer.cuda._backend import lib

    L = lib()
    rc = L.gsr_project_forward(ctypes.c_int(4), None, None, ctypes.c_float(1), None, None, None,
                               ctypes.c_float(1), ctypes.c_float(1), ctypes.c_float(0), ctypes.c_float(0),
                               ctypes.c_uint(8), ctypes.c_uint(8), ctypes.c_uint(17), ctypes.c_float(0.01),
                               None, None, None, None, None, None, None, None)
    assert rc == -1 and b"block_width" in L.gsr_last_error()
    rc = L.gsr_sh_forward(ctypes.c_uint(4), ctypes.c_uint(5), ctypes.c_uint(0), None, None, None, None)
    assert rc == -1 and b"degree" in L.gsr_last_error()
    rc = L.gsr_rasterize_forward(ctypes.c_int(3), ctypes.c_int(1), ctypes.c_uint(16), ctypes.c_uint(32),
                                 ctypes.c_uint(16), None, None, None, None, None, None, None, None, None, None,
                                 ctypes.c_int(0), None)
    assert rc == -1 and b"tile bounds" in L.gsr_last_error()
    # zero-sized work is a no-op, not an error
    assert L.gsr_sh_forward(ctypes.c_uint(0), ctypes.c_uint(3), ctypes.c_uint(3), None, None, None, None) == 0
    assert L.gsr_sort_workspace_bytes(ctypes.c_int(0)) == 0


def test_package_surface_matches_reference():
    import rasterizer
    import rasterizer.cuda as C
    from rasterizer._torch_impl import quat_to_rotmat
    from rasterizer.project_gaussians import project_gaussians  # noqa: F401
    from rasterizer.rasterize import rasterize_gaussians  # noqa: F401
    from rasterizer.sh import num_sh_bases, spherical_harmonics  # noqa: F401

    assert rasterizer.__version__ == "0.1.2"
    for name in ("project_gaussians", "rasterize_gaussians", "spherical_harmonics", "bin_and_sort_gaussians",
                 "compute_cumulative_intersects", "compute_cov2d_bounds", "get_tile_bin_edges",
                 "map_gaussian_to_intersects", "ProjectGaussians", "RasterizeGaussians", "BinAndSortGaussians",
                 "ComputeCumulativeIntersects", "ComputeCov2dBounds", "GetTileBinEdges",
                 "MapGaussiansToIntersects", "SphericalHarmonics", "NDRasterizeGaussians"):
        assert name in rasterizer.__all__ and hasattr(rasterizer, name)
    # the 11 functions of the reference's pybind module (ext.cpp:6-17)
    for fn in ("nd_rasterize_forward", "nd_rasterize_backward", "rasterize_forward", "rasterize_backward",
               "project_gaussians_forward", "project_gaussians_backward", "compute_sh_forward",
               "compute_sh_backward", "compute_cov2d_bounds", "map_gaussian_to_intersects",
               "get_tile_bin_edges"):
        assert callable(getattr(C, fn))
    import torch

    q = torch.tensor([[2.0, 0.0, 0.0, 0.0], [0.5, 0.5, 0.5, 0.5]])
    R = quat_to_rotmat(q)
    assert torch.allclose(R[0], torch.eye(3)) and torch.allclose(R[1] @ R[1].T, torch.eye(3), atol=1e-6)


def test_cpu_tensors_are_rejected_loudly():
    import torch

    import rasterizer.cuda as C
    from rasterizer import project_gaussians

    with pytest.raises(RuntimeError, match="CUDA tensor"):
        C.compute_sh_forward(2, 0, 0, torch.zeros(2, 3), torch.zeros(2, 1, 3))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        project_gaussians(torch.zeros(2, 3), torch.ones(2, 3), 1, torch.ones(2, 4), torch.eye(4)[:3],
                          torch.eye(4), 1, 1, 0, 0, 8, 8, 16)


def test_oracle_finite_differences():
    """The oracle's hand-written VJPs agree with central differences of its
    forward, on the parts of the path that are smooth: the projection (all
    outputs, all parameters) and the compositing w.r.t. colours (exactly linear)
    and opacities.  (Compositing w.r.t. geometry is cut off at alpha < 1/255 and
    T <= 1e-4, so finite differences see boundary terms the analytic gradient --
    like the reference's -- does not have; that part is pinned by torch.autograd
    through the reference implementation in test_oracle_golden.py.)"""
    from harness import scene as S
    from oracle import oracle as O

    cam = S.make_camera(64, 48, yaw=0.15, pitch=-0.1, roll=0.05, trans=(0.1, -0.2, 0.3))
    n = 10
    sc = S.make_scene(n, cam, sh_degree=0, seed=5, scale_lo=0.05, scale_hi=0.3, z_lo=1.5, z_hi=4.0)
    rng = np.random.default_rng(0)
    wx, wd, wc, wk = (rng.standard_normal(s) for s in ((n, 2), (n,), (n, 3), (n,)))

    def proj(means, scales, quats):
        return O.project_gaussians_forward(n, means, scales, 1.0, quats, cam.viewmat[:3], cam.projmat, cam.fx,
                                           cam.fy, cam.cx, cam.cy, 48, 64, 16, 0.01)

    def ploss(means, scales, quats):
        cov3d, xys, depths, radii, conics, comp, tiles = proj(means, scales, quats)
        return float((xys.astype(np.float64) * wx).sum() + (depths.astype(np.float64) * wd).sum() +
                     (conics.astype(np.float64) * wc).sum() + (comp.astype(np.float64) * wk).sum())

    # unit quaternions: the kernel's quaternion VJP assumes |q| = 1 and the FD
    # below perturbs along the tangent of the sphere only
    cov3d, xys, depths, radii, conics, comp, tiles = proj(sc["means3d"], sc["scales"], sc["quats"])
    vis = radii > 0
    assert vis.sum() >= 6
    _, _, vmean, vscale, vquat = O.project_gaussians_backward(
        n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat, cam.fx, cam.fy, cam.cx,
        cam.cy, 48, 64, cov3d, radii, conics, comp, wx.astype(np.float32), wd.astype(np.float32),
        wc.astype(np.float32), wk.astype(np.float32))
    checked = 0
    for i in np.nonzero(vis)[0]:
        for nm, arr, grad in (("means3d", sc["means3d"], vmean), ("scales", sc["scales"], vscale)):
            for j in range(3):
                h = 1e-3 * max(0.1, abs(float(arr[i, j])))
                p, m = arr.astype(np.float64).copy(), arr.astype(np.float64).copy()
                p[i, j] += h
                m[i, j] -= h
                kw = dict(means=sc["means3d"], scales=sc["scales"], quats=sc["quats"])
                key = "means" if nm == "means3d" else "scales"
                fp = ploss(**{**kw, key: p.astype(np.float32)})
                fm = ploss(**{**kw, key: m.astype(np.float32)})
                fd = (fp - fm) / (float(np.float32(p[i, j])) - float(np.float32(m[i, j])))
                an = float(grad[i, j])
                assert abs(fd - an) <= 2e-2 * max(abs(fd), abs(an)) + 2e-2, (nm, i, j, fd, an)
                checked += 1
        # quaternion: directional derivative along a tangent direction
        q = sc["quats"][i].astype(np.float64)
        d = rng.standard_normal(4)
        d -= q * (d @ q)
        d /= np.linalg.norm(d)
        h = 1e-3
        qp, qm = sc["quats"].astype(np.float64).copy(), sc["quats"].astype(np.float64).copy()
        qp[i] = (q + h * d) / np.linalg.norm(q + h * d)
        qm[i] = (q - h * d) / np.linalg.norm(q - h * d)
        fd = (ploss(sc["means3d"], sc["scales"], qp.astype(np.float32)) -
              ploss(sc["means3d"], sc["scales"], qm.astype(np.float32))) / (2 * h)
        an = float(vquat[i].astype(np.float64) @ d)
        assert abs(fd - an) <= 3e-2 * max(abs(fd), abs(an)) + 3e-2, ("quat", i, fd, an)
        checked += 1
    assert checked >= 40

    # compositing: colours (linear) and opacities
    colors = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    w = rng.uniform(-1, 1, (48, 64, 3))
    u = rng.uniform(-1, 1, (48, 64))
    tb = (4, 3, 1)
    I, cum = O.compute_cumulative_intersects(tiles)
    _, _, ks, vs, bins = O.bin_and_sort_gaussians(n, I, xys, depths, radii, cum, tb, 16)
    opac = np.clip(sc["opacities"], 0.2, 0.8)

    def rloss(col, op):
        out, Ts, idx = O.rasterize_forward(tb, (16, 16, 1), (64, 48, 1), vs, bins, xys, conics, col, op, bg)
        return float((out.astype(np.float64) * w).sum() + ((1 - Ts.astype(np.float64)) * u).sum()), Ts, idx

    _, Ts, idx = rloss(colors, opac)
    vxy, vconic, vcol, vop = O.rasterize_backward(48, 64, 16, vs, bins, xys, conics, colors, opac, bg, Ts, idx,
                                                  w.astype(np.float32), u.astype(np.float32))
    for i in np.nonzero(vis)[0][:6]:
        for c in range(3):
            p, m = colors.copy(), colors.copy()
            p[i, c] += 0.25
            m[i, c] -= 0.25
            fd = (rloss(p, opac)[0] - rloss(m, opac)[0]) / 0.5
            assert abs(fd - float(vcol[i, c])) <= 1e-3 * max(1.0, abs(fd)), ("color", i, c, fd, vcol[i, c])
        p, m = opac.copy(), opac.copy()
        p[i, 0] += 2e-3
        m[i, 0] -= 2e-3
        fd = (rloss(colors, p)[0] - rloss(colors, m)[0]) / (float(p[i, 0]) - float(m[i, 0]))
        an = float(vop[i, 0])
        assert abs(fd - an) <= 0.1 * max(abs(fd), abs(an)) + 0.05, ("opacity", i, fd, an)


def test_header_is_plain_c(tmp_path):
    """include/gsraster.h is the drop-in boundary: it must compile as C99 on its own
    (extern "C" guards, no C++ or torch types)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    src.write_text('#include "gsraster.h"\nint main(void) { gsr_adam_tensor t; (void)t; return GSR_OK; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src),
                           "-o", str(tmp_path / "use_header.o")])


def test_descriptor_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of the descriptor structs (`gsr_raster_desc`, `gsr_view_desc`, `gsr_view_grads`) must have
    the C compiler's size and field offsets: a mismatch would hand the library shifted pointers."""
    import ctypes as C
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    import rasterizer.cuda as RC
    from gs_fused.render import _ViewDesc, _ViewGrads

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = (("gsr_raster_desc", RC._RasterDesc), ("gsr_view_desc", _ViewDesc), ("gsr_view_grads", _ViewGrads))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gsraster.h"', "int main(void) {"]
    for cname, cls in structs:
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    want = {}
    for line in out.splitlines():
        s, f, v = line.split()
        want[(s, f)] = int(v)
    for cname, cls in structs:
        assert C.sizeof(cls) == want[(cname, "size")], cname
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == want[(cname, fname)], (cname, fname)
