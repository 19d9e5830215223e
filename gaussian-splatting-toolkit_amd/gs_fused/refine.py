"""Refinement of a Gaussian model -- densify (split / duplicate), cull, opacity reset --
with the Adam state carried along, as one compaction on the GPU.

Mirror of ``GaussianSplattingModel.refinement_after``
(gs_toolkit/models/vanilla_gs.py:381-497; ``split_gaussians`` :540-592,
``dup_gaussians`` :594-603, ``cull_gaussians`` :499-538, ``dup_in_optim`` :303-337,
``remove_from_optim`` :282-301) over ``gsr_refine_plan`` / ``gsr_refine_apply``
(include/gsraster.h, csrc/refine.hip): the decisions are taken in one pass over the
per-Gaussian scalars, then every parameter tensor and both Adam moments are
streamed ONCE into their final layout (the reference concatenates, then culls:
~60 launches and two copies of everything).  One host read-back per call -- the
row counts, needed to allocate the outputs; the reference has one per mask.

The order of the output rows is the reference's: surviving originals, split
children sample by sample, duplicates.  fp32 CUDA tensors only; no CPU fallback.
"""
import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from rasterizer.cuda import _call, _check, _lib, _ptr, _stream

PARAM_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
_f32 = torch.float32
KIND_COPY, KIND_LOG_SCALES, KIND_MEANS, KIND_MOMENT = 0, 1, 2, 3
MAX_TENSORS = 24


@dataclass
class RefineConfig:
    """The refinement fields of ``GaussianSplattingModelConfig`` (vanilla_gs.py:40-106),
    reference defaults."""
    warmup_length: int = 500
    refine_every: int = 100
    cull_alpha_thresh: float = 0.1
    cull_scale_thresh: float = 0.5
    continue_cull_post_densification: bool = True
    reset_alpha_every: int = 30
    densify_grad_thresh: float = 0.0002
    densify_size_thresh: float = 0.01
    n_split_samples: int = 2
    cull_screen_size: float = 0.15
    split_screen_size: float = 0.05
    stop_screen_size_at: int = 4000
    stop_split_at: int = 10_000


class _Config(C.Structure):
    _fields_ = [("densify_grad_thresh", C.c_float), ("densify_size_thresh", C.c_float),
                ("split_screen_size", C.c_float), ("cull_alpha_thresh", C.c_float),
                ("cull_scale_thresh", C.c_float), ("cull_screen_size", C.c_float), ("half_max_dim", C.c_float),
                ("n_split_samples", C.c_int), ("densify", C.c_int), ("split_by_screen_size", C.c_int),
                ("cull_big", C.c_int), ("cull_by_screen_size", C.c_int)]


class _RefineTensor(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("width", C.c_int), ("kind", C.c_int)]


def refinement_branch(cfg: RefineConfig, step: int, num_train_data: int) -> Tuple[str, bool]:
    """Which branch ``refinement_after`` takes at `step` -> (``"none"`` / ``"densify"`` /
    ``"cull"``, reset the opacities?).  Host arithmetic only (vanilla_gs.py:383-395,
    459-473)."""
    if step <= cfg.warmup_length:
        return "none", False
    reset_interval = cfg.reset_alpha_every * cfg.refine_every
    if step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every:
        branch = "densify"
    elif step >= cfg.stop_split_at and cfg.continue_cull_post_densification:
        branch = "cull"
    else:
        branch = "none"
    reset = step < cfg.stop_split_at and step % reset_interval == cfg.refine_every
    return branch, reset


@torch.no_grad()
def refine_gaussians(
    params: Dict[str, Tensor],
    moments: Optional[Dict[str, Tuple[Tensor, Tensor]]],
    stats: Optional[Tuple[Tensor, Tensor, Tensor]],
    cfg: RefineConfig,
    step: int,
    num_train_data: int,
    max_dim: int,
    samples: Optional[Tensor] = None,
    seed: int = 0,
):
    """One ``refinement_after(step)``.

    params:  the six raw parameter tensors (``PARAM_NAMES``; log-scales, unnormalised
             quats, opacity logits), ``[N, ...]`` fp32 CUDA.
    moments: name -> ``(exp_avg, exp_avg_sq)`` of that parameter's Adam state (same
             shapes), or None / missing names for parameters without state yet.
    stats:   ``(xys_grad_norm [N] f32, vis_counts [N] i32, max_2dsize [N] f32)`` as
             accumulated by ``densify_stats_``; needed on densification steps and
             whenever a screen-size rule is on.
    max_dim: ``max(W, H)`` of the last rendered view (vanilla_gs.py:404-408).
    samples: optional ``[n_split_samples * n_splits, 3]`` N(0,1) draws in the
             reference's ``torch.randn`` layout; default: generated in the kernel
             from ``seed`` (counter-based, identical on every replica).

    Returns ``(new_params, new_moments, info)``.  When nothing changes the input
    tensors themselves are returned (no copy).  ``info``: branch, counts, and whether
    the opacities were reset (in which case their Adam moments were zeroed, :470-489).
    """
    branch, reset = refinement_branch(cfg, step, num_train_data)
    info = {"branch": branch, "opacity_reset": reset, "n_in": params["means"].shape[0],
            "n_out": params["means"].shape[0], "kept": None, "split": 0, "dup": 0}
    new_params, new_moments = dict(params), (None if moments is None else dict(moments))
    if branch != "none":
        new_params, new_moments = _compact(params, moments, stats, cfg, step, branch == "densify", max_dim,
                                           samples, seed, info)
    if reset:
        # reset value = twice the cull threshold, in logit space (:471-478)
        lim = torch.logit(torch.tensor(cfg.cull_alpha_thresh * 2.0, dtype=_f32)).item()
        new_params["opacities"].clamp_(max=lim)  # `self.opacities.data = clamp(...)`: same tensor, new values
        if new_moments is not None and new_moments.get("opacities") is not None:
            for t in new_moments["opacities"]:
                t.zero_()
    return new_params, new_moments, info


def _compact(params, moments, stats, cfg, step, densify, max_dim, samples, seed, info):
    n = params["means"].shape[0]
    dev = params["means"].device
    S = int(cfg.n_split_samples)
    for k in PARAM_NAMES:
        _check(params[k], k, _f32)
        if params[k].shape[0] != n:
            raise ValueError(f"{k} has {params[k].shape[0]} rows, means has {n}")
    if params["scales"].shape != (n, 3) or params["quats"].shape != (n, 4) or params["means"].shape != (n, 3) \
            or params["opacities"].numel() != n:
        raise ValueError("expected means [N,3], scales [N,3], quats [N,4], opacities [N,1]")
    split_by_screen = densify and step < cfg.stop_screen_size_at
    cull_big = step > cfg.refine_every * cfg.reset_alpha_every
    cull_by_screen = cull_big and step < cfg.stop_screen_size_at
    gn = vc = m2 = None
    if stats is not None:
        gn, vc, m2 = stats
        _check(gn, "xys_grad_norm", _f32), _check(vc, "vis_counts", torch.int32), _check(m2, "max_2dsize", _f32)
        if gn.numel() != n or vc.numel() != n or m2.numel() != n:
            raise ValueError("the densification statistics must have N elements")
    if densify and stats is None:
        raise ValueError("densification needs the statistics of after_train")  # the reference asserts (:398-402)
    if (split_by_screen or cull_by_screen) and m2 is None:
        raise ValueError("the screen-size rules need max_2dsize")
    if n == 0:
        return dict(params), (None if moments is None else dict(moments))
    c = _Config(cfg.densify_grad_thresh, cfg.densify_size_thresh, cfg.split_screen_size, cfg.cull_alpha_thresh,
                cfg.cull_scale_thresh, cfg.cull_screen_size, 0.5 * float(max_dim), S, int(densify),
                int(split_by_screen), int(cull_big), int(cull_by_screen))
    opt = lambda t: None if t is None else _ptr(t)
    with torch.cuda.device(dev):
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        offsets = torch.empty((n, 4), dtype=torch.int32, device=dev)
        counts = torch.empty(4, dtype=torch.int32, device=dev)
        wbytes = int(_lib().gsr_refine_workspace_bytes(C.c_int(n)))
        work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        _call("gsr_refine_plan", C.c_int(n), _ptr(params["scales"]), _ptr(params["opacities"]), opt(gn), opt(vc),
              opt(m2), C.byref(c), _ptr(flags), _ptr(offsets), _ptr(counts), _ptr(work), C.c_size_t(wbytes),
              _stream(dev))
        k0, ks, kd, nsplit = (int(v) for v in counts.tolist())  # the one host read-back
        n_out = k0 + S * ks + kd
        info.update(n_out=n_out, kept=k0, split=ks, dup=kd, split_sources=nsplit)
        if k0 == n and ks == 0 and kd == 0:
            return dict(params), (None if moments is None else dict(moments))  # nothing moves
        if samples is not None:
            samples = _check(samples.contiguous(), "samples", _f32)
            if samples.numel() != 3 * S * nsplit:
                raise ValueError(f"samples must be [{S * nsplit}, 3] (n_split_samples * number of split Gaussians)")
        items = []  # (in, out, width, kind)
        new_params, new_moments = {}, (None if moments is None else {})
        kinds = {"means": KIND_MEANS, "scales": KIND_LOG_SCALES}
        for k in PARAM_NAMES:
            t = params[k]
            out = torch.empty((n_out,) + tuple(t.shape[1:]), dtype=_f32, device=dev)
            new_params[k] = out
            w = t.numel() // n
            items.append((t, out, w, kinds.get(k, KIND_COPY)))
            if moments is not None and moments.get(k) is not None:
                outs = []
                for m in moments[k]:
                    m = _check(m if m.is_contiguous() else m.contiguous(), f"Adam state of {k}", _f32)
                    if m.shape != t.shape:
                        raise ValueError(f"Adam state of {k} has shape {tuple(m.shape)}, parameter {tuple(t.shape)}")
                    o = torch.empty_like(out)
                    outs.append(o)
                    items.append((m, o, w, KIND_MOMENT))
                new_moments[k] = tuple(outs)
        # (zero-width rows -- `features_rest` of an SH-degree-0 model is [N, 0, 3] -- have nothing to move:
        #  their outputs above already have the new number of rows)
        items = [it for it in items if it[2] > 0]
        for i in range(0, len(items), MAX_TENSORS):
            chunk = items[i:i + MAX_TENSORS]
            arr = (_RefineTensor * len(chunk))()
            for j, (a, b, w, kind) in enumerate(chunk):
                arr[j] = _RefineTensor(a.data_ptr(), b.data_ptr(), w, kind)
            _call("gsr_refine_apply", C.c_int(n), C.c_int(S), _ptr(flags), _ptr(offsets), _ptr(counts),
                  _ptr(params["scales"]), _ptr(params["quats"]), opt(samples), C.c_ulonglong(int(seed) & (2**64 - 1)),
                  C.c_int(len(chunk)), arr, _stream(dev))
    return new_params, new_moments


def adam_moments(optimizer: torch.optim.Optimizer, params: Dict[str, Tensor]) -> Dict[str, Tuple[Tensor, Tensor]]:
    """``{name: (exp_avg, exp_avg_sq)}`` for the named parameters that have Adam state."""
    out = {}
    for k, p in params.items():
        st = optimizer.state.get(p)
        if st and "exp_avg" in st:
            out[k] = (st["exp_avg"], st["exp_avg_sq"])
    return out


def swap_parameters(optimizer: torch.optim.Optimizer, old: Dict[str, Tensor], new: Dict[str, Tensor],
                    new_moments: Optional[Dict[str, Tuple[Tensor, Tensor]]]) -> None:
    """Point `optimizer` (``torch.optim.Adam`` or ``FusedAdam``; any number of param
    groups) at the new parameter objects, carrying ``step`` over and installing the
    compacted moments -- what remove_from_optim / dup_in_optim do per group
    (vanilla_gs.py:282-337)."""
    ident = {id(old[k]): k for k in old}
    for group in optimizer.param_groups:
        for i, p in enumerate(group["params"]):
            k = ident.get(id(p))
            if k is None or new[k] is p:
                continue
            st = optimizer.state.pop(p, None)
            group["params"][i] = new[k]
            if st:
                if new_moments is not None and new_moments.get(k) is not None:
                    st["exp_avg"], st["exp_avg_sq"] = new_moments[k]
                optimizer.state[new[k]] = st
