"""numpy restatement of the toolkit's Gaussian refinement (densify / cull / split /
duplicate with Adam-state surgery).

TEST INFRASTRUCTURE ONLY (see ``oracle/oracle.py``): the checker of
``csrc/refine.hip`` / ``gs_fused/refine.py``; the product never imports it.

Follows ``GaussianSplattingModel`` in ``gs_toolkit/models/vanilla_gs.py``:
``after_train`` :344-372, ``refinement_after`` :381-497, ``cull_gaussians``
:499-538, ``split_gaussians`` :540-592, ``dup_gaussians`` :594-603,
``dup_in_optim`` :303-337, ``remove_from_optim`` :282-301.  Pinned on the
reference's own methods: ``tests/golden/make_golden_refine.py`` extracts those
methods from the reference source at generation time, runs them on torch CPU
tensors with ``torch.optim.Adam`` objects and commits inputs/outputs as
``tests/golden/refine.npz``; ``tests/test_refine.py`` checks this file against it.

All arithmetic is float32 like the torch ops it restates.  The random samples of
``split_gaussians`` (``torch.randn((samps * n_splits, 3))``, :543) are an input
(``samples``, the reference's layout: row ``j * n_splits + rank(i)``), or come from
the counter-based generator below (Philox4x32-10 keyed on ``(seed, i, j)`` +
Box-Muller), which is what the HIP kernel uses when no samples are handed in: the
values then depend on the Gaussian's index only, not on how many others split, and
are identical on every data-parallel replica.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

PARAM_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
f32 = np.float32


@dataclass
class RefineConfig:
    """The fields of ``GaussianSplattingModelConfig`` (vanilla_gs.py:40-106) that
    refinement reads, with the reference's defaults."""
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


SIZE_FAC = f32(1.6)  # vanilla_gs.py:564


# ---- counter-based normal samples (shared definition with csrc/refine.hip) ----------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """counter [...,4] uint32, key [...,2] uint32 -> [...,4] uint32 (Salmon et al.,
    SC'11, the Random123 constants)."""
    c = [counter[..., k].astype(np.uint64) for k in range(4)]
    k0 = key[..., 0].astype(np.uint32)
    k1 = key[..., 1].astype(np.uint32)
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c = [(hi1 ^ c[1] ^ k0.astype(np.uint64)) & _MASK, lo1, (hi0 ^ c[3] ^ k1.astype(np.uint64)) & _MASK, lo0]
        with np.errstate(over="ignore"):
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def split_normals(seed: int, gaussian_index: np.ndarray, sample: int) -> np.ndarray:
    """Three N(0,1) floats for (Gaussian i, split sample j): Philox block with counter
    (i, j, 0, 0) and key (seed low, seed high); u = ((x >> 9) + 0.5) * 2^-23 in (0,1);
    Box-Muller pairs (x0,x1) -> z0, z1 and (x2,x3) -> z2."""
    i = np.asarray(gaussian_index, dtype=np.uint32)
    ctr = np.stack([i, np.full_like(i, sample), np.zeros_like(i), np.zeros_like(i)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), i.shape + (2,))
    x = philox4x32_10(ctr, key)
    u = ((x >> np.uint32(9)).astype(f32) + f32(0.5)) * f32(2.0 ** -23)  # exact in fp32, inside (0, 1)
    r0 = np.sqrt(f32(-2) * np.log(u[..., 0]))
    r1 = np.sqrt(f32(-2) * np.log(u[..., 2]))
    two_pi = f32(6.283185307179586)
    z0 = r0 * np.cos(two_pi * u[..., 1])
    z1 = r0 * np.sin(two_pi * u[..., 1])
    z2 = r1 * np.cos(two_pi * u[..., 3])
    return np.stack([z0, z1, z2], axis=-1).astype(f32)


# ---- helpers ----------------------------------------------------------------
def _sigmoid(x):
    return (f32(1) / (f32(1) + np.exp(-x.astype(f32)))).astype(f32)


def quat_to_rotmat(q: np.ndarray) -> np.ndarray:
    """rasterizer/_torch_impl.py::quat_to_rotmat (normalises, (w,x,y,z))."""
    q = q / np.linalg.norm(q, axis=-1, keepdims=True).astype(f32)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    one, two = f32(1), f32(2)
    R = np.stack([
        one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y),
        two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x),
        two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)], axis=-1)
    return R.reshape(q.shape[:-1] + (3, 3)).astype(f32)


def update_stats(stats, v_xys: np.ndarray, radii: np.ndarray, max_dim: int):
    """after_train (vanilla_gs.py:344-372).  stats = (xys_grad_norm, vis_counts,
    max_2dsize) or None; returns the new triple (float32, float32, float32)."""
    visible = radii > 0
    grads = np.sqrt((v_xys.astype(f32) ** 2).sum(-1)).astype(f32)
    if stats is None:
        gn, vc = grads.copy(), np.ones_like(grads)
        m2 = np.zeros(len(radii), f32)
    else:
        gn, vc, m2 = (a.copy() for a in stats)
        vc[visible] += 1
        gn[visible] = grads[visible] + gn[visible]
    m2[visible] = np.maximum(m2[visible], radii[visible].astype(f32) / f32(max_dim))
    return gn, vc, m2


def refine(params: Dict[str, np.ndarray], moments: Optional[Dict[str, Tuple[np.ndarray, np.ndarray]]],
           stats, cfg: RefineConfig, step: int, num_train_data: int, max_dim: int,
           samples: Optional[np.ndarray] = None, seed: int = 0):
    """One call of ``refinement_after`` at `step`.

    params: name -> float32 [N,...]; moments: name -> (exp_avg, exp_avg_sq) or None
    (optimizer without state); stats: the triple of `update_stats` (needed on
    densification steps, and for the screen-size cull).  Returns
    ``(params, moments, info)``; info holds the masks (`splits`, `dups`, `culls`
    over the concatenated set) and `samples` actually used."""
    p = {k: np.array(v, dtype=f32, copy=True) for k, v in params.items()}
    mom = None if moments is None else {k: (np.array(a, f32, copy=True), np.array(b, f32, copy=True))
                                        for k, (a, b) in moments.items()}
    info = {"splits": None, "dups": None, "culls": None, "samples": None, "opacity_reset": False}
    if step <= cfg.warmup_length:  # :383
        return p, mom, info
    n = p["means"].shape[0]
    reset_interval = cfg.reset_alpha_every * cfg.refine_every
    do_densification = step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every
    m2 = None if stats is None else stats[2].astype(f32)

    def cull(extra):  # cull_gaussians :499-538
        culls = _sigmoid(p["opacities"]).reshape(-1) < f32(cfg.cull_alpha_thresh)
        if extra is not None:
            culls = culls | extra
        if step > cfg.refine_every * cfg.reset_alpha_every:
            toobigs = np.exp(p["scales"]).max(-1) > f32(cfg.cull_scale_thresh)
            if step < cfg.stop_screen_size_at:
                toobigs = toobigs | (m2 > f32(cfg.cull_screen_size))
            culls = culls | toobigs
        for k in p:
            p[k] = p[k][~culls]
        return culls

    culls = None
    if do_densification:
        gn, vc, _ = stats
        with np.errstate(divide="ignore", invalid="ignore"):
            avg = ((gn.astype(f32) / vc.astype(f32)) * f32(0.5) * f32(max_dim)).astype(f32)  # :404-408
        high = avg > f32(cfg.densify_grad_thresh)
        splits = np.exp(p["scales"]).max(-1) > f32(cfg.densify_size_thresh)
        if step < cfg.stop_screen_size_at:
            splits = splits | (m2 > f32(cfg.split_screen_size))
        splits = splits & high
        S = cfg.n_split_samples
        idx = np.nonzero(splits)[0]
        ns = len(idx)
        # split_gaussians :540-592
        if samples is None:
            z = np.concatenate([split_normals(seed, idx, j) for j in range(S)], 0) if ns else np.zeros((0, 3), f32)
        else:
            z = np.asarray(samples, f32).reshape(S * ns, 3)
        rep = lambda a: np.concatenate([a] * S, 0)
        scaled = (np.exp(rep(p["scales"][splits])) * z).astype(f32)
        R = quat_to_rotmat(rep(p["quats"][splits])) if ns else np.zeros((0, 3, 3), f32)
        rotated = np.einsum("nij,nj->ni", R, scaled).astype(f32)
        child = {
            "means": (rotated + rep(p["means"][splits])).astype(f32),
            "features_dc": rep(p["features_dc"][splits]),
            "features_rest": rep(p["features_rest"][splits]),
            "opacities": rep(p["opacities"][splits]),
            "scales": rep(np.log(np.exp(p["scales"][splits]) / SIZE_FAC).astype(f32)),
            "quats": rep(p["quats"][splits]),
        }
        p["scales"][splits] = np.log(np.exp(p["scales"][splits]) / SIZE_FAC).astype(f32)  # in place, :568
        # duplicates are chosen AFTER the in-place shrink (:433-438): a split Gaussian whose
        # shrunk scale falls under the threshold is duplicated too (with the shrunk scale)
        dups = (np.exp(p["scales"]).max(-1) <= f32(cfg.densify_size_thresh)) & high
        nd = int(dups.sum())
        for k in p:
            p[k] = np.concatenate([p[k], child[k], p[k][dups]], 0)
        m2 = np.concatenate([m2, np.zeros(S * ns + nd, f32)])
        if mom is not None:  # dup_in_optim :303-337 (zeros for the new rows)
            for k, (a, b) in mom.items():
                z0 = np.zeros((S * ns + nd,) + a.shape[1:], f32)
                mom[k] = (np.concatenate([a, z0], 0), np.concatenate([b, z0], 0))
        splits_mask = np.concatenate([splits, np.zeros(S * ns + nd, bool)])
        culls = cull(splits_mask)
        info.update(splits=splits, dups=dups, samples=z)
    elif step >= cfg.stop_split_at and cfg.continue_cull_post_densification:
        culls = cull(None)
    if culls is not None:
        if mom is not None:  # remove_from_optim :282-301
            for k, (a, b) in mom.items():
                mom[k] = (a[~culls], b[~culls])
        info["culls"] = culls
    if step < cfg.stop_split_at and step % reset_interval == cfg.refine_every:  # :470-489
        reset_value = f32(cfg.cull_alpha_thresh * 2.0)
        lim = float(np.log(reset_value / (f32(1) - reset_value)))  # torch.logit in fp32, .item()
        p["opacities"] = np.minimum(p["opacities"], f32(lim))
        if mom is not None and "opacities" in mom:
            a, b = mom["opacities"]
            mom["opacities"] = (np.zeros_like(a), np.zeros_like(b))
        info["opacity_reset"] = True
    return p, mom, info
