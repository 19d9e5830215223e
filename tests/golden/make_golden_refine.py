#!/usr/bin/env python3
"""Generate tests/golden/refine.npz from the REFERENCE's own refinement code.

Runs ONLY in the build container: it reads `/root/reference/gs_toolkit/models/
vanilla_gs.py`, which does not exist on the GPU box.  `gs_toolkit` cannot be imported
here (tyro, jaxtyping, viser, pytorch_msssim ... are absent), so the methods of
`GaussianSplattingModel` that make up refinement -- `after_train`,
`refinement_after`, `cull_gaussians`, `split_gaussians`, `dup_gaussians`,
`dup_in_optim`, `dup_in_all_optim`, `remove_from_optim`, `remove_from_all_optim`,
`get_gaussian_param_groups` and the parameter properties -- are lifted out of the
class with `ast` AT GENERATION TIME and executed unmodified on a bare host object
(torch CPU tensors, real `torch.optim.Adam` objects).  Nothing of the reference's
source is stored: the committed `.npz` holds inputs and the outputs those methods
produced.

`torch.randn` inside `split_gaussians` is wrapped so that the samples it drew are
recorded as an input of the case (the generator stream of torch is not something
another implementation can reproduce).

    python tests/golden/make_golden_refine.py
"""
import ast
import os
import sys
import types
from typing import Dict, List, Optional

import numpy as np
import torch
from torch.nn import Parameter

HERE = os.path.dirname(os.path.abspath(__file__))
REF_MODEL = "/root/reference/gs_toolkit/models/vanilla_gs.py"
REF_RAST = "/root/reference/gs_toolkit/gs_components"
METHODS = {
    "after_train", "refinement_after", "cull_gaussians", "split_gaussians", "dup_gaussians", "dup_in_optim",
    "dup_in_all_optim", "remove_from_optim", "remove_from_all_optim", "get_gaussian_param_groups", "num_points",
    "means", "scales", "quats", "features_dc", "features_rest", "opacities",
}
NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


def _reference_class():
    """A class holding the reference's refinement methods, compiled from its source."""
    import tempfile

    shim = tempfile.mkdtemp(prefix="jaxtyping_shim_")
    with open(os.path.join(shim, "jaxtyping.py"), "w") as f:
        f.write("class _T:\n    def __class_getitem__(cls, item):\n        return cls\n"
                "class Float(_T): pass\nclass Int(_T): pass\n")
    sys.path.insert(0, shim)
    sys.path.insert(0, REF_RAST)
    import rasterizer._torch_impl as ti

    assert ti.__file__.startswith(REF_RAST), ti.__file__
    tree = ast.parse(open(REF_MODEL).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianSplattingModel")
    body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS]
    found = {n.name for n in body}
    assert found == METHODS, METHODS - found
    mod = ast.Module(body=[ast.ClassDef(name="RefModel", bases=[], keywords=[], body=body, decorator_list=[])],
                     type_ignores=[])
    ast.fix_missing_locations(mod)
    recorded = []

    class _TorchProxy(types.ModuleType):
        def __getattr__(self, k):
            return getattr(torch, k)

    tp = _TorchProxy("torch")

    def randn(*a, **k):
        z = torch.randn(*a, **k)
        recorded.append(z.clone())
        return z

    tp.randn = randn
    ns = {"torch": tp, "quat_to_rotmat": ti.quat_to_rotmat, "Optional": Optional, "Dict": Dict, "List": List,
          "Parameter": Parameter, "Optimizers": object, "np": np}
    exec(compile(mod, REF_MODEL, "exec"), ns)
    return ns["RefModel"], recorded


def make_case(rng, Ref, recorded, n, step, cfg_over, num_train_data, size, after_steps):
    """Random model + Adam state + `after_steps` calls of after_train, then one
    refinement_after(step).  Returns dict of arrays."""
    cfg = dict(warmup_length=500, refine_every=100, cull_alpha_thresh=0.1, cull_scale_thresh=0.5,
               continue_cull_post_densification=True, reset_alpha_every=30, densify_grad_thresh=0.0002,
               densify_size_thresh=0.01, n_split_samples=2, cull_screen_size=0.15, split_screen_size=0.05,
               stop_screen_size_at=4000, stop_split_at=10_000)
    cfg.update(cfg_over)
    f = np.float32
    raw = {
        "means": rng.standard_normal((n, 3)).astype(f),
        # log-scales: both sides of densify_size_thresh (0.01), a few above cull_scale_thresh (0.5)
        "scales": np.log(np.exp(rng.uniform(np.log(0.002), np.log(0.9), (n, 3)))).astype(f),
        "quats": rng.standard_normal((n, 4)).astype(f),
        "features_dc": rng.standard_normal((n, 3)).astype(f),
        "features_rest": rng.standard_normal((n, 15, 3)).astype(f),
        "opacities": rng.uniform(-4.0, 4.0, (n, 1)).astype(f),  # sigmoid: 0.018 .. 0.98
    }
    m = Ref.__new__(Ref)
    m.config = types.SimpleNamespace(**cfg)
    m.device = torch.device("cpu")
    m.gauss_params = torch.nn.ParameterDict({k: Parameter(torch.from_numpy(v.copy())) for k, v in raw.items()})
    m.num_train_data = num_train_data
    m.xys_grad_norm = m.vis_counts = m.max_2Dsize = None
    m.last_size = size  # (H, W)
    opts = types.SimpleNamespace(optimizers={k: torch.optim.Adam([m.gauss_params[k]], lr=1e-3, eps=1e-15)
                                             for k in NAMES})
    # give Adam a non-trivial state: two steps with random gradients
    for _ in range(2):
        for k in NAMES:
            m.gauss_params[k].grad = torch.from_numpy(rng.standard_normal(raw[k].shape).astype(f) * 0.01)
            opts.optimizers[k].step()
    p_in = {k: m.gauss_params[k].detach().numpy().copy() for k in NAMES}
    mom_in = {k: tuple(opts.optimizers[k].state[m.gauss_params[k]][s].numpy().copy()
                       for s in ("exp_avg", "exp_avg_sq")) for k in NAMES}
    # densification statistics through the reference's after_train
    views = []
    m.step = step
    for _ in range(after_steps):
        radii = (rng.uniform(0, 1, n) < 0.7) * rng.integers(1, int(0.25 * max(size)), n)
        vxy = (rng.standard_normal((n, 2)) * 10 ** rng.uniform(-8, -4, (n, 1))).astype(f)
        vxy[radii == 0] = 0
        m.radii = torch.from_numpy(radii.astype(np.int32))
        m.xys = types.SimpleNamespace(grad=torch.from_numpy(vxy))
        m.after_train(step)
        views.append((vxy, radii.astype(np.int32)))
    stats = None
    if m.xys_grad_norm is not None:
        stats = (m.xys_grad_norm.numpy().copy(), m.vis_counts.numpy().copy(), m.max_2Dsize.numpy().copy())
    del recorded[:]
    m.refinement_after(opts, step)
    samples = recorded[0].numpy() if recorded else np.zeros((0, 3), f)
    out = {"step": np.int64(step), "num_train_data": np.int64(num_train_data), "size": np.array(size, np.int64),
           "samples": samples, "n_views": np.int64(len(views))}
    for k, v in cfg.items():
        out["cfg_" + k] = np.array(v)
    for i, (vxy, radii) in enumerate(views):
        out[f"view{i}_vxy"], out[f"view{i}_radii"] = vxy, radii
    if stats is not None:
        out["stat_gn"], out["stat_vc"], out["stat_m2"] = stats
    for k in NAMES:
        out["in_" + k] = p_in[k]
        out["in_m_" + k], out["in_v_" + k] = mom_in[k]
        newp = m.gauss_params[k]
        out["out_" + k] = newp.detach().numpy().copy()
        st = opts.optimizers[k].state[opts.optimizers[k].param_groups[0]["params"][0]]
        out["out_m_" + k], out["out_v_" + k] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy()
        assert out["out_m_" + k].shape == out["out_" + k].shape, (k, out["out_m_" + k].shape, out["out_" + k].shape)
    return out


def main():
    torch.manual_seed(1234)
    rng = np.random.default_rng(99)
    Ref, recorded = _reference_class()
    cases = {
        # name: (n, step, cfg overrides, num_train_data, (H, W), after_train calls)
        "warmup": (120, 400, {}, 20, (90, 160), 3),                      # step <= warmup: nothing
        "densify_screen": (240, 700, {}, 20, (90, 160), 5),              # densify, screen-size rules on, no big cull
        "densify_bigcull": (240, 3700, {}, 20, (90, 160), 5),            # step > 3000: scale + screen-size cull
        "densify_late": (240, 4700, {}, 20, (90, 160), 5),               # step >= stop_screen_size_at
        "densify_3samples": (200, 4500, {"n_split_samples": 3}, 20, (120, 100), 4),
        "no_densify_window": (150, 3050, {}, 20, (90, 160), 3),          # step % 3000 <= num_train_data + 100
        "opacity_reset": (150, 3100, {}, 20, (90, 160), 3),              # step % 3000 == refine_every
        "cull_only": (200, 10_000, {}, 20, (90, 160), 0),                # post densification cull
        "cull_off": (150, 10_100, {"continue_cull_post_densification": False}, 20, (90, 160), 0),
        "densify_lowthresh": (300, 5300, {"cull_alpha_thresh": 0.005, "densify_grad_thresh": 0.0006}, 40,
                              (64, 64), 6),
    }
    blob = {}
    for name, (n, step, over, ntd, size, calls) in cases.items():
        c = make_case(rng, Ref, recorded, n, step, over, ntd, size, calls)
        n_out = c["out_means"].shape[0]
        print(f"{name:20s} step {step:6d}  N {n} -> {n_out}  samples {c['samples'].shape[0]}")
        for k, v in c.items():
            blob[f"{name}/{k}"] = v
    blob["cases"] = np.array(list(cases.keys()))
    np.savez_compressed(os.path.join(HERE, "refine.npz"), **blob)
    print("wrote", os.path.join(HERE, "refine.npz"), os.path.getsize(os.path.join(HERE, "refine.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
