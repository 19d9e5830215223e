"""A minimal trainer with the toolkit's call pattern, for
BASELINE configs 3/4: `gs-train gaussian-splatting` on a synthetic scene, one
GPU or per-view data parallel.

What is reproduced from the reference (so that the rasterizer sees exactly the
calls the real models make):
  * parameters and activations of `GaussianSplattingModel`
    (gs_toolkit/models/vanilla_gs.py:128-174, 765-820): means, log-scales,
    unnormalised quats, logit opacities, features_dc [N,3], features_rest
    [N,15,3]; `exp`, normalise, `sigmoid`, SH degree warm-up
    `min(step // sh_degree_interval, sh_degree)`, `clamp(SH + 0.5, min=0)`;
  * the loss `(1 - lambda) L1 + lambda (1 - SSIM)` with lambda = 0.2
    (vanilla_gs.py:900-947; SSIM restated in plain torch, 11x11 Gaussian window);
  * one Adam optimiser per parameter group with the learning rates of
    gs_toolkit/configs/method_configs.py:98-132;
  * `xys.retain_grad()` and the densification statistics of `after_train`
    (vanilla_gs.py:344-372), and -- with `TrainConfig.densify` -- the refinement
    schedule of `refinement_after` (:381-497) every `refine_every` iterations:
    split / duplicate / cull / opacity reset with the Adam state carried along
    (`gs_fused.refine_gaussians`, one compaction launch).  Under data parallelism
    the statistics are all-reduced first and the split samples come from a
    counter-based generator keyed on (seed, step, Gaussian index), so every replica
    takes the same decisions and stays bit-identical.
The data side (cameras on a sphere, ground truth rendered by this rasterizer
from a hidden "true" scene) replaces the toolkit's datamanager.
"""
import math
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import cogs_losses
from . import scene as S
from .parallel import GradientExchange, allreduce_densify_stats, view_for_rank
from .pipeline import CameraTensors, render_view

PARAM_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")

LRS = {"means": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacities": 0.05,
       "scales": 0.005, "quats": 0.001}
SH_C0 = 0.28209479177387814


def orbit_cameras(n_views: int, width: int, height: int, radius: float = 6.0, fov_x_deg: float = 60.0):
    """Cameras on a circle around the origin, all looking at it (rasterizer
    convention: x right, y down, z forward)."""
    cams = []
    for i in range(n_views):
        th = 2 * math.pi * i / n_views
        eye = np.array([radius * math.sin(th), 0.6 * math.sin(2 * th), -radius * math.cos(th)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(np.array([0.0, -1.0, 0.0]), fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])  # world -> camera rows
        cam = S.make_camera(width, height, fov_x_deg)
        V = np.eye(4, dtype=np.float32)
        V[:3, :3] = R.astype(np.float32)
        V[:3, 3] = (-R @ eye).astype(np.float32)
        fovx = math.radians(fov_x_deg)
        fovy = 2.0 * math.atan(height / (2.0 * cam.fy))
        P = S.projection_matrix(0.001, 1000.0, fovx, fovy) @ V
        cams.append(S.Camera(width, height, cam.fx, cam.fy, cam.cx, cam.cy, V, P.astype(np.float32)))
    return cams


def rescale_camera(cam, d: int):
    """`camera.rescale_output_resolution(1 / d)` (gs_toolkit/cameras/cameras.py:1176-1213: fx, fy, cx, cy scaled,
    width and height scaled and truncated to integers) followed by what `get_outputs` derives from the rescaled
    camera (vanilla_gs.py:736-742: the fields of view from the NEW width / fx, the projection matrix from those)."""
    if d == 1:
        return cam
    f = 1.0 / d
    fx, fy, cx, cy = cam.fx * f, cam.fy * f, cam.cx * f, cam.cy * f
    W, H = int(cam.width * f), int(cam.height * f)
    fovx, fovy = 2.0 * math.atan(W / (2.0 * fx)), 2.0 * math.atan(H / (2.0 * fy))
    P = S.projection_matrix(0.001, 1000.0, fovx, fovy) @ cam.viewmat
    return S.Camera(W, H, fx, fy, cx, cy, cam.viewmat, P.astype(np.float32))


def downscale_factor(step: int, num_downscales: int, resolution_schedule: int) -> int:
    """`_get_downscale_factor` while training (vanilla_gs.py:646-657): 2^max(num_downscales - step // schedule, 0)."""
    return 2 ** max(num_downscales - step // max(resolution_schedule, 1), 0)


def downscale_image(image: torch.Tensor, d: int) -> torch.Tensor:
    """`_downscale_if_required` (vanilla_gs.py:659-670): `TF.resize(image.permute(2, 0, 1), [H // d, W // d],
    antialias=None)` -- for tensors that is bilinear interpolation without antialiasing."""
    if d <= 1:
        return image
    newsize = [image.shape[0] // d, image.shape[1] // d]
    return F.interpolate(image.permute(2, 0, 1)[None], size=newsize, mode="bilinear", align_corners=False,
                         antialias=False)[0].permute(1, 2, 0)


def downscale_depth(depth: torch.Tensor, d: int) -> torch.Tensor:
    """`DepthGSModel._downscale_if_required` for a 2-D image (depth_gs.py:161-176): resized as [1,H,W]."""
    if d <= 1:
        return depth
    return F.interpolate(depth[None, None], size=[depth.shape[0] // d, depth.shape[1] // d], mode="bilinear",
                         align_corners=False, antialias=False)[0, 0]


def composite_with_background(image: torch.Tensor, background: torch.Tensor) -> torch.Tensor:
    """What the models do with a ground-truth image that carries an alpha channel (vanilla_gs.py:870-881): straight
    colour over the step's background, `a * rgb + (1 - a) * background`; an RGB image is returned as it is."""
    if image.shape[-1] != 4:
        return image
    rgb, a = image[..., :3], image[..., 3:4]
    return a * rgb + (1.0 - a) * background


def blob_scene(n: int, seed: int, sh_degree: int = 3, extent: float = 1.5, scale_lo=0.01, scale_hi=0.06,
               kind: str = "ball", tex_cell: float = 0.04, objects=(48, 0.18, 0.45)):
    """Raw (pre-activation) Gaussians around the origin.  kind="ball": semi-transparent
    Gaussians filling a ball (a smooth volume).  kind="shell": opaque, small, flat-ish
    Gaussians on three nested textured spheres -- surfaces with detail at the pixel scale,
    the kind of scene densification exists for."""
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((n, 3))
    p /= np.linalg.norm(p, axis=-1, keepdims=True)
    K = S.num_sh_bases(sh_degree)
    f32 = np.float32
    if kind == "objects":
        # a nerfstudio-style object scene: `n_obj` opaque spheres of different sizes scattered in a
        # ball (silhouettes against the background, mutual occlusion), each tiled by small FLAT
        # Gaussians aligned with the surface and textured with a high-contrast checker whose cells
        # are a few Gaussians wide -- detail a sparse start cannot represent without densifying
        n_obj, r_lo, r_hi = int(objects[0]), float(objects[1]), float(objects[2])
        centres = rng.standard_normal((n_obj, 3))
        centres *= (extent * rng.uniform(0.15, 1.0, (n_obj, 1)) ** (1 / 3)) / np.linalg.norm(centres, axis=-1, keepdims=True)
        radii = rng.uniform(r_lo, r_hi, n_obj)
        which = rng.choice(n_obj, n, p=radii ** 2 / np.sum(radii ** 2))  # by area
        r = radii[which][:, None]
        means = centres[which] + p * (r + 0.002 * rng.standard_normal((n, 1)))
        base = rng.uniform(0.15, 0.85, (n_obj, 3))[which]
        other = rng.uniform(0.0, 1.0, (n_obj, 3))[which]
        cell = tex_cell / r[:, 0]  # checker cell size in radians: `tex_cell` scene units on every sphere
        u = np.arctan2(p[:, 0], p[:, 2]) * np.maximum(np.sqrt(1 - p[:, 1] ** 2), 0.2)
        v = np.arcsin(np.clip(p[:, 1], -1, 1))
        chk = ((np.floor(u / cell) + np.floor(v / cell)) % 2)[:, None]
        colour = np.where(chk > 0, base, other) * (0.85 + 0.15 * rng.uniform(0, 1, (n, 1)))
        dc = (np.clip(colour, 0, 1) - 0.5) / SH_C0
        s_t = rng.uniform(math.log(scale_lo), math.log(scale_hi), (n, 2))
        scales = np.concatenate([s_t, s_t.min(axis=1, keepdims=True) + math.log(0.15)], axis=1)  # thin along the normal
        # quaternion (w, x, y, z) turning the local z axis into the outward normal p
        quats = np.stack([1.0 + p[:, 2], -p[:, 1], p[:, 0], np.zeros(n)], axis=-1)
        quats[np.linalg.norm(quats, axis=-1) < 1e-6] = (0.0, 1.0, 0.0, 0.0)
        quats /= np.linalg.norm(quats, axis=-1, keepdims=True)
        return {
            "means": means.astype(f32), "scales": scales.astype(f32), "quats": quats.astype(f32),
            "opacities": rng.uniform(2.0, 5.0, (n, 1)).astype(f32), "features_dc": dc.astype(f32),
            "features_rest": (rng.standard_normal((n, K - 1, 3)) * 0.02).astype(f32),
        }
    if kind == "shell":
        radius = np.array([0.7, 1.1, 1.5])[rng.integers(0, 3, n)][:, None]
        means = p * (radius + 0.004 * rng.standard_normal((n, 1)))
        # texture: colour varies smoothly over the sphere plus per-Gaussian contrast
        tex = 0.5 + 0.5 * np.sin(9.0 * p @ rng.standard_normal((3, 3)) + 5.0 * radius)
        dc = (0.7 * tex + 0.3 * rng.uniform(0, 1, (n, 3)) - 0.5) / SH_C0
        scales = rng.uniform(math.log(scale_lo), math.log(scale_hi), (n, 3))
        opac = rng.uniform(1.0, 4.0, (n, 1))
    else:
        means = p * (extent * rng.uniform(0, 1, (n, 1)) ** (1 / 3))
        dc = (rng.uniform(0, 1, (n, 3)) - 0.5) / SH_C0
        scales = np.log(np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), (n, 3))))
        opac = rng.uniform(-1.0, 2.0, (n, 1))  # logits
    return {
        "means": means.astype(f32),
        "scales": scales.astype(f32),
        "quats": rng.standard_normal((n, 4)).astype(f32),
        "opacities": opac.astype(f32),
        "features_dc": dc.astype(f32),
        "features_rest": (rng.standard_normal((n, K - 1, 3)) * 0.05).astype(f32),
    }


def knn_mean_distance(points: np.ndarray, k: int = 3) -> np.ndarray:
    """Mean distance to the k nearest neighbours, the reference's initial scale
    (`k_nearest_sklearn`, vanilla_gs.py:136-140, 260-280: k + 1 neighbours, self dropped)."""
    from sklearn.neighbors import NearestNeighbors

    d, _ = NearestNeighbors(n_neighbors=k + 1, algorithm="auto", metric="euclidean").fit(points).kneighbors(points)
    return d[:, 1:].astype(np.float32).mean(axis=-1)


def random_quats(n: int, rng) -> np.ndarray:
    """`random_quat_tensor` (gs_toolkit/utils/comms.py:69-84): uniform on the unit 3-sphere."""
    u, v, w = rng.uniform(size=n), rng.uniform(size=n), rng.uniform(size=n)
    return np.stack([np.sqrt(1 - u) * np.sin(2 * math.pi * v), np.sqrt(1 - u) * np.cos(2 * math.pi * v),
                     np.sqrt(u) * np.sin(2 * math.pi * w), np.sqrt(u) * np.cos(2 * math.pi * w)], -1).astype(np.float32)


def seed_model(truth: Dict[str, np.ndarray], n_seed: int, kind: str, seed: int, sh_degree: int = 3,
               sfm_noise: float = 0.01, random_scale: float = 3.4):
    """The model's start as `GaussianSplattingModel.populate_modules` builds it
    (vanilla_gs.py:128-174) -- NOT a copy of the truth:
      kind="sfm":    a sparse point cloud with 8-bit colours, standing in for the COLMAP points
                     a nerfstudio-style dataset ships: `n_seed` surface points of the hidden scene
                     with `sfm_noise` of triangulation error; colour = the point's view-independent
                     colour.  means = the points, features_dc = RGB2SH(colour);
      kind="random": `random_init=True`: means uniform in a cube of edge `random_scale`
                     (the reference's 10 is for its unit-normalised real scenes; 3.4 encloses
                     this scene's radius-1.5 shells), features_dc = rand.
    Both: log-scales = log(mean distance to the 3 nearest seeds) on all three axes, random unit
    quaternions, opacity logit(0.1), higher SH bands zero."""
    rng = np.random.default_rng(seed)
    K = S.num_sh_bases(sh_degree)
    f32 = np.float32
    if kind == "sfm":
        pick = np.sort(rng.choice(truth["means"].shape[0], n_seed, replace=False))
        means = truth["means"][pick] + rng.standard_normal((n_seed, 3)).astype(f32) * f32(sfm_noise)
        rgb8 = np.round(np.clip(truth["features_dc"][pick] * SH_C0 + 0.5, 0, 1) * 255.0)
        dc = ((rgb8 / 255.0) - 0.5) / SH_C0
    elif kind == "random":
        means = (rng.uniform(size=(n_seed, 3)) - 0.5) * random_scale
        dc = rng.uniform(size=(n_seed, 3))
    else:
        raise ValueError(f"unknown seed kind {kind!r}")
    means = means.astype(f32)
    avg = np.maximum(knn_mean_distance(means, 3), 1e-7)
    return {
        "means": means,
        "scales": np.repeat(np.log(avg)[:, None], 3, axis=1).astype(f32),
        "quats": random_quats(n_seed, rng),
        "opacities": np.full((n_seed, 1), math.log(0.1 / 0.9), f32),
        "features_dc": dc.astype(f32),
        "features_rest": np.zeros((n_seed, K - 1, 3), f32),
    }


def means_lr(step: int, lr_init: float = 1.6e-4, lr_final: float = 1.6e-6, max_steps: int = 30000) -> float:
    """`ExponentialDecayScheduler` without warm-up, the schedule of the "means" group
    (configs/method_configs.py:98-104, engine/schedulers.py:94-135)."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class GaussianParams(torch.nn.Module):
    split_sh = True  # evaluate SH from (features_dc, features_rest) without torch.cat
    fused_activations = True  # exp / normalise / sigmoid / view directions in one launch (gs_fused)

    def __init__(self, raw: Dict[str, np.ndarray], device):
        super().__init__()
        self.gauss = torch.nn.ParameterDict(
            {k: torch.nn.Parameter(torch.from_numpy(v).to(device)) for k, v in raw.items()})

    @property
    def num_points(self):
        return self.gauss["means"].shape[0]

    def param_list(self) -> List[torch.nn.Parameter]:
        return [self.gauss[k] for k in PARAM_NAMES]

    def replace(self, new: Dict[str, torch.Tensor]) -> None:
        """Install refined tensors as the model's parameters (vanilla_gs.py:440-447, 532)."""
        for k in PARAM_NAMES:
            if new[k] is not self.gauss[k]:
                self.gauss[k] = torch.nn.Parameter(new[k])

    def render(self, cam: CameraTensors, background, sh_degree_to_use: int, render_depth=False,
               retain_xys_grad=False, clamp_rgb=True, sh_exchange=None, caller_syncs=False, fused_depth=False,
               normalise_depth=True):
        g = self.gauss
        if self.split_sh and g["features_dc"].is_cuda and g["features_rest"].shape[1] in (3, 8, 15):
            coeffs = (g["features_dc"], g["features_rest"])  # gs_fused.spherical_harmonics_split
        else:
            coeffs = torch.cat((g["features_dc"][:, None, :], g["features_rest"]), dim=1)
        if self.fused_activations and g["means"].is_cuda:
            from gs_fused import activate_gaussians

            scales, quats, opac, dirs = activate_gaussians(g["means"], g["scales"], g["quats"], g["opacities"],
                                                           cam.campos)
        else:
            scales = torch.exp(g["scales"])
            quats = g["quats"] / g["quats"].norm(dim=-1, keepdim=True)
            opac, dirs = torch.sigmoid(g["opacities"]), None
        return render_view(g["means"], scales, quats, opac, coeffs, cam, background, sh_degree_to_use,
                           render_depth=render_depth, retain_xys_grad=retain_xys_grad, viewdirs=dirs,
                           clamp_rgb=clamp_rgb, caller_syncs=caller_syncs, fused_depth=fused_depth,
                           normalise_depth=normalise_depth,
                           sh_exchange=None if sh_exchange is None else (
                               sh_exchange, ("features_dc", "features_rest"), (g["features_dc"], g["features_rest"])))


def _gauss_window(size=11, sigma=1.5, device="cpu"):
    x = torch.arange(size, dtype=torch.float32, device=device) - size // 2
    g = torch.exp(-(x * x) / (2 * sigma * sigma))
    return g / g.sum()


def ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """Mean SSIM of two [H,W,3] images in [0,1] (11x11 Gaussian window, sigma 1.5)."""
    a = img1.permute(2, 0, 1)[None]
    b = img2.permute(2, 0, 1)[None]
    w = _gauss_window(device=a.device)
    C = a.shape[1]
    kx = w.view(1, 1, 1, -1).repeat(C, 1, 1, 1)
    ky = w.view(1, 1, -1, 1).repeat(C, 1, 1, 1)
    blur = lambda t: F.conv2d(F.conv2d(t, kx, groups=C), ky, groups=C)
    mu1, mu2 = blur(a), blur(b)
    s11 = blur(a * a) - mu1 * mu1
    s22 = blur(b * b) - mu2 * mu2
    s12 = blur(a * b) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    return m.mean()


def cogs_main_loss(pred: torch.Tensor, target: torch.Tensor, ssim_lambda: float) -> torch.Tensor:
    """`DepthGSModel.get_loss_dict`'s photometric terms op by op (depth_gs.py:441-462), quirk included: the SSIM is
    computed and then DROPPED -- `+self.config.ssim_lambda * simloss` stands on a line of its own (:447-448) -- so
    `main_loss` = (1 - ssim_lambda) * L1; `scale_reg` is tensor(0.) with the default config and the trainer adds
    the dict's values up (engine/trainer.py:497)."""
    Ll1 = torch.abs(target - pred).mean()
    simloss = 1 - ssim(target, pred)
    main_loss = (1 - ssim_lambda) * Ll1
    +ssim_lambda * simloss  # noqa: B018  (an expression statement, as in the source)
    scale_reg = torch.tensor(0.0).to(pred.device)
    return main_loss + scale_reg


def cogs_depth_l1(pred_depth: torch.Tensor, gt_depth: torch.Tensor) -> torch.Tensor:
    """`depth_l1` of depth_gs.py:531-538: |gt * (gt > 0) - pred * (gt > 0)|.mean() over ALL pixels, pred [H,W,1]
    squeezed; added to the loss with weight 1 (`depth_lambda` is declared twice in the config, :85 and :115, and
    read nowhere)."""
    depth_nonzero = gt_depth > 0
    pred = pred_depth.squeeze(-1)
    return torch.abs(gt_depth * depth_nonzero - pred * depth_nonzero).mean()


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(-10.0 * torch.log10(((a - b) ** 2).mean().clamp_min(1e-12)))


@dataclass
class TrainConfig:
    num_gaussians: int = 100_000
    width: int = 640
    height: int = 360
    num_views: int = 24
    iters: int = 300
    sh_degree: int = 3
    sh_degree_interval: int = 1000
    ssim_lambda: float = 0.2
    seed: int = 0
    log_every: int = 0
    eval_views: int = 4
    fused_loss: bool = True   # gs_fused.l1_ssim_loss (2 HIP kernels) instead of the torch-op SSIM
    fused_adam: bool = True   # gs_fused.FusedAdam: all six parameter groups in one HIP launch
    torch_fused_adam: bool = False  # (A/B) torch's own fused multi-tensor Adam instead
    split_sh: bool = True     # gs_fused.spherical_harmonics_split instead of torch.cat + spherical_harmonics
    fused_activations: bool = True  # gs_fused.activate_gaussians instead of exp / normalise / sigmoid / viewdirs
    # refinement (vanilla_gs.py:381-497).  Off: N stays fixed and iterations are comparable.
    densify: bool = False
    init_gaussians: Optional[int] = None  # the model starts from this many (coarser) Gaussians; default: all
    refine: Optional[object] = None       # gs_fused.RefineConfig; default: the reference's values
    refine_seed: int = 20240807           # broadcast by construction: the same on every rank
    # gs_fused.render_gaussians: the whole view as ONE autograd node (activations, projection, SH,
    # binning, compositing; densification statistics from its backward) instead of the op-by-op
    # call sequence of the models; use_graph: render -> loss -> backward replayed as one HIP graph
    fused_render: bool = False
    use_graph: bool = False
    # checkpoints in the toolkit's layout (harness/checkpoint.py; trainer.py:404-476)
    checkpoint_dir: Optional[str] = None
    save_every: int = 0                   # steps_per_save; 0 = never
    resume_from: Optional[str] = None     # a .ckpt file or a directory (latest step)
    scene: str = "ball"                   # blob_scene kind
    # where the model starts: "perturbed" = the hidden scene with noise on the means and washed-out
    # colour (optionally a subset, init_gaussians); "sfm" / "random" = the reference's own
    # initialisations from a sparse point cloud / uniformly random points (seed_model)
    init: str = "perturbed"
    means_lr_schedule: bool = False       # exponential decay of the "means" learning rate (method_configs.py:98-104)
    # data parallel only: reduce-scatter -> Adam on this rank's rows -> all-gather (parallel.ShardedAdam)
    # instead of all-reduce + Adam over every row on every rank
    sharded_adam: bool = False
    # data parallel only: "views" = the SH gradient is formed on every rank from the ranks' gathered 12-byte colour
    # cotangents (GradientExchange `sh_views`: 413 -> 161 B per Gaussian on the wire at 8 ranks), "dense" = every
    # gradient is all-reduced.  "views" needs the hook-driven exchange and the separate-ops render path.
    sh_exchange: str = "views"
    phase_every: int = 0                  # HIP events around render / loss / backward / optimizer on every k-th iteration
    phase_series: bool = False            # also return every sample (`phase_ms_series`: step, resolution divisor, 4 phases)
    scene_scale: tuple = (0.01, 0.06)     # range of the truth's Gaussian scales
    tex_cell: float = 0.04                # scene "objects": checker cell size in scene units
    scene_objects: tuple = (48, 0.18, 0.45)  # scene "objects": number of spheres, radius range
    cam_radius: float = 6.0               # radius of the camera orbit
    scene_extent: float = 1.5             # radius of the ball the scene fills
    # the reference's coarse-to-fine schedule (vanilla_gs.py:48-53, 646-669): the first `resolution_schedule`
    # iterations render at 1 / 2^num_downscales of the resolution, the next at half of that factor, ... (reference
    # defaults: 2 and 2000, i.e. 480x270 -> 960x540 -> 1920x1080 at steps 2000 and 4000; 0 here = full size from step 0)
    num_downscales: int = 0
    resolution_schedule: int = 2000
    # "random": a fresh `torch.rand(3)` background every training step (vanilla_gs.py:50, 688-690 -- the reference's
    # default), the ground truth then carries an alpha channel and is composited over it (:870-881, a
    # nerfstudio-synthetic / blender-style RGBA dataset); "fixed": scene.BACKGROUND, RGB ground truth
    background_color: str = "fixed"
    # block the host where the unchanged models do (pipeline.render_view `caller_syncs`): False, True (the two
    # read-backs of get_outputs, vanilla_gs.py:784,811) or "camera" (plus the intrinsics' .item() calls)
    caller_syncs: object = False
    # run the data-parallel machinery (gradient exchange, sharded Adam, statistics all-reduce) on a process group of
    # ONE rank too: every collective then goes through the backend as an identity -- how the RCCL code path is
    # exercised on a single-GPU box (tests/test_gpu_nccl.py)
    force_exchange: bool = False
    export_ply: Optional[str] = None      # write the trained model as `gs-export gaussian-splat` does (gs_io/ply.py)
    # random background only: the step's target as ONE kernel.  The reference downscales the RGBA ground truth and
    # composites it over the step's background every step (vanilla_gs.py:659-670, 870-881: a resize + four
    # elementwise kernels over the image).  The resized image does not depend on the step: alpha_ds * rgb_ds and
    # 1 - alpha_ds (both formed from the RESIZED RGBA, in the reference's order -- resizing does not commute with
    # the product) are cached per view and resolution and the step runs one `addcmul` with its background.
    # False: the reference's per-step sequence.
    fused_target: bool = True
    # "gaussian-splatting": GaussianSplattingModel (vanilla_gs.py).  "co-gs": DepthGSModel (depth_gs.py) -- BASELINE
    # config 5: the depth image is rasterised on the TRAINING path (`output_depth_during_training = True`, :99,
    # :345-363), the photometric loss is (1 - ssim_lambda) * L1 ONLY (:445-448: the `+ ssim_lambda * simloss` line is
    # an expression statement -- the SSIM is computed and dropped; followed as written), and from
    # `step > depth_loss_start_iteration` (:119-120, :472-476) the L1 between the rendered depth and the ground
    # truth's on the pixels with gt > 0 is ADDED UNWEIGHTED (:532-538; the trainer sums the dict,
    # engine/trainer.py:497 -- `depth_lambda` is never read).  Refinement, optimisers and schedules are the base
    # model's.  Set num_downscales = 0 and background_color = "random" for the reference's co-gs defaults (:51, :49).
    model: str = "gaussian-splatting"
    depth_loss_start_iteration: int = 6000
    use_depth_loss: bool = True
    # co-gs: RGB and depth from ONE compositing pass each way (gs_fused.rasterize_gaussians_rgbd) and the depth
    # normalisation + masked L1 in gs_fused.depth_l1_loss; False: the models' two `rasterize_gaussians` calls and
    # their torch ops (depth_gs.py:330-363, 531-538)
    fused_depth: bool = True
    # co-gs: the OPTIONAL loss terms of `DepthGSModelConfig` (depth_gs.py:93-139), every one off by default as in the
    # reference; restated in harness/cogs_losses.py (plain torch ops, outside the rasterizer's hot path).  With
    # `use_est_depth` the depth branch is the monocular-depth one (:477-531: local Pearson / scaled log-depth / TV
    # instead of `depth_l1`); `use_depth_regularization` needs OpenCV's Canny and raises.
    use_scale_regularization: bool = False
    max_gauss_ratio: float = 10.0
    use_sparse_loss: bool = False
    sparse_lambda: float = 0.1
    use_est_depth: bool = False
    use_pearson_depth: bool = False
    local_patch_size: int = 128
    depth_loss_stop_iteration: int = 25_000
    use_scaled_est_depth: bool = False
    use_depth_regularization: bool = False
    using_tv_loss: bool = False


def quantise_depth_mm(depth: torch.Tensor) -> torch.Tensor:
    """A ground-truth depth image as the toolkit's dataset delivers it: a 16-bit PNG in millimetres read as
    `astype("float32") / 1000.0` (data/datasets/base_dataset.py:123-125); 0 = no measurement."""
    return torch.round(depth.clamp(0.0, 65.535) * 1000.0) / 1000.0


def _depth_segments_record(device):
    """What the compositing kernels do on tile grids too small to fill the chip (rasterizer.cuda.depth_segments)."""
    if device.type != "cuda":
        return None
    import rasterizer.cuda as _C

    segs, grid, least, fwd = _C._segment_knobs()[:4]
    return {"runs": segs, "runs_forward": min(segs, fwd) if fwd > 0 else segs, "tile_grids_up_to": grid,
            "lists_longer_than": least,
            "what": "on such grids the list of every split tile is cut into runs composited by their own waves "
                    "(gsr_rasterize_forward_seg / _backward_seg): results equal the single walk's to rounding"}


def _sh_views_backward_autograd():
    """CPU stand-in of gs_fused.sh_backward_views: the same sum over views through the autograd of whatever
    `harness.pipeline.spherical_harmonics` is (SH is linear in the coefficients)."""
    import harness.pipeline as HP

    def fn(degree, deg_use, means, campos_all, v_all, scale, split):
        n, K = means.shape[0], (degree + 1) ** 2
        coeffs = torch.zeros((n, K, 3), dtype=means.dtype, device=means.device, requires_grad=True)
        total = torch.zeros((n, K, 3), dtype=means.dtype, device=means.device)
        for r in range(campos_all.shape[0]):
            d = means - campos_all[r]
            d = d / d.norm(dim=-1, keepdim=True)
            (g,) = torch.autograd.grad(HP.spherical_harmonics(deg_use, d, coeffs), coeffs, v_all[r].reshape(n, 3))
            total += g
        total *= scale
        return (total[:, 0, :].contiguous(), total[:, 1:, :].contiguous()) if split else total

    return fn


def train(cfg: TrainConfig, device, rank: int = 0, world: int = 1) -> Dict:
    """Fit a perturbed copy of a hidden scene to its own renders.  Returns timing
    and quality numbers; every rank ends with identical parameters."""
    dp = world > 1 or (cfg.force_exchange and dist.is_available() and dist.is_initialized())
    cams_np = orbit_cameras(cfg.num_views, cfg.width, cfg.height, radius=cfg.cam_radius)
    cams = [CameraTensors.from_numpy(c, device) for c in cams_np]
    # the coarse-to-fine schedule's cameras, one set per downscale factor
    factors = sorted({downscale_factor(s_, cfg.num_downscales, cfg.resolution_schedule)
                      for s_ in range(0, max(cfg.iters, 1), max(min(cfg.resolution_schedule, cfg.iters), 1))} | {1})
    cams_by_d = {d: (cams if d == 1 else [CameraTensors.from_numpy(rescale_camera(c, d), device) for c in cams_np])
                 for d in factors}
    bg = torch.tensor(S.BACKGROUND, device=device)
    random_bg = cfg.background_color == "random"
    if cfg.background_color not in ("random", "fixed"):
        raise ValueError(f"unknown background_color {cfg.background_color!r}")
    # (the reference seeds every rank differently, scripts/train.py:54: the backgrounds differ per rank, as there)
    bg_gen = torch.Generator(device=device).manual_seed(cfg.seed + 977 * (rank + 1)) if random_bg else None

    if cfg.model not in ("gaussian-splatting", "co-gs"):
        raise ValueError(f"unknown model {cfg.model!r}")
    cogs = cfg.model == "co-gs"
    if cogs and (cfg.fused_render or cfg.use_graph):
        raise ValueError("co-gs runs through the separate ops (fused_depth = one RGB + depth compositing pass)")
    mk = lambda: blob_scene(cfg.num_gaussians, seed=cfg.seed, sh_degree=cfg.sh_degree, kind=cfg.scene,
                            scale_lo=cfg.scene_scale[0], scale_hi=cfg.scene_scale[1], tex_cell=cfg.tex_cell,
                            objects=cfg.scene_objects, extent=cfg.scene_extent)
    truth = GaussianParams(mk(), device)
    with torch.no_grad():
        if random_bg:
            # RGBA ground truth with straight (un-premultiplied) colour, as a blender-style dataset stores it:
            # rendered over black, C = sum c_i a_i T_i and A = 1 - T, colour = C / A
            zero = torch.zeros(3, device=device)
            gt_rgba, gt = [], []
            for c in cams:
                o = truth.render(c, zero, cfg.sh_degree, clamp_rgb=False)
                a = o["alpha"]
                gt_rgba.append(torch.cat((torch.where(a > 0, o["rgb"] / a.clamp_min(1e-12), torch.zeros_like(o["rgb"]))
                                          .clamp(0, 1), a), dim=-1))
                gt.append(composite_with_background(gt_rgba[-1], bg))  # evaluation: over the fixed background
        else:
            gt_rgba = None
            gt = [truth.render(c, bg, cfg.sh_degree)["rgb"] for c in cams]
        gt_depth = None
        if cogs:
            # the sensor's depth image: z-depth of the hidden scene where it covers the pixel, 0 (= no measurement,
            # masked out by `gt_depth > 0`) elsewhere, in millimetre steps like the dataset's 16-bit PNGs
            gt_depth = []
            for c in cams:
                o = truth.render(c, torch.zeros(3, device=device), cfg.sh_degree, render_depth=True,
                                 fused_depth=device.type == "cuda")
                gt_depth.append(quantise_depth_mm(torch.where(o["alpha"] > 0.5, o["depth"], torch.zeros_like(o["depth"]))
                                                  [..., 0]).contiguous())

    raw = mk()
    rng = np.random.default_rng(cfg.seed + 1)
    if cfg.init in ("sfm", "random"):
        raw = seed_model(raw, cfg.init_gaussians or 50_000, cfg.init, cfg.seed + 1, cfg.sh_degree)
    else:
        # the model starts from the truth with perturbed geometry / washed-out colour
        raw["means"] += rng.standard_normal(raw["means"].shape).astype(np.float32) * 0.01
        raw["features_dc"] *= 0.3
        raw["features_rest"] *= 0.0
        raw["opacities"] -= 0.5
    if cfg.init == "perturbed" and cfg.init_gaussians is not None and cfg.init_gaussians < cfg.num_gaussians:
        # a coarser start for densification to refine: a subset, each Gaussian standing in for
        # num/init of the truth's (scales grown by the cube root of that ratio)
        keep = np.sort(rng.choice(cfg.num_gaussians, cfg.init_gaussians, replace=False))
        raw = {k: np.ascontiguousarray(v[keep]) for k, v in raw.items()}
        raw["scales"] += np.float32(math.log(cfg.num_gaussians / cfg.init_gaussians) / 3.0)
    model = GaussianParams(raw, device)
    model.split_sh = truth.split_sh = cfg.split_sh
    model.fused_activations = truth.fused_activations = cfg.fused_activations
    if cfg.fused_adam and device.type == "cuda":
        # one optimiser, six parameter groups with the reference's learning rates
        groups = [{"params": [model.gauss[k]], "lr": lr} for k, lr in LRS.items()]
        if cfg.torch_fused_adam:
            optims = {"all": torch.optim.Adam(groups, eps=1e-15, fused=True)}
        else:
            from gs_fused import FusedAdam

            optims = {"all": FusedAdam(groups, eps=1e-15)}
    else:
        optims = {k: torch.optim.Adam([model.gauss[k]], lr=lr, eps=1e-15) for k, lr in LRS.items()}
    sharded = None
    if cfg.sharded_adam and dp:
        from .parallel import ShardedAdam

        if cfg.fused_adam and device.type == "cuda":
            from gs_fused import FusedAdam

            make = lambda groups: FusedAdam(groups, eps=1e-15)
        else:
            make = lambda groups: torch.optim.Adam(groups, eps=1e-15)
        sharded = ShardedAdam({k: model.gauss[k] for k in PARAM_NAMES}, LRS, make, force=cfg.force_exchange)
        optims = {}
    fused_clamp = False
    if cfg.fused_loss and device.type == "cuda":
        from gs_fused import l1_ssim_loss

        # the clamp of the rendered image at 1 (vanilla_gs.py:857) is folded into the loss kernels
        fused_clamp = True
        loss_fn = lambda pred, target: l1_ssim_loss(pred, target, cfg.ssim_lambda, clamp_pred=fused_clamp)
    else:
        loss_fn = lambda pred, target: ((1 - cfg.ssim_lambda) * (pred - target).abs().mean()
                                        + cfg.ssim_lambda * (1 - ssim(pred, target)))

    depth_loss_fn = None
    if cogs:
        w_l1 = 1.0 - cfg.ssim_lambda
        if cfg.fused_loss and device.type == "cuda":
            from gs_fused import depth_l1_loss, l1_loss

            loss_fn = lambda pred, target: l1_loss(pred, target, w_l1, clamp_pred=fused_clamp)
        else:
            loss_fn = lambda pred, target: cogs_main_loss(pred, target, cfg.ssim_lambda)
        if cfg.fused_depth and cfg.fused_loss and device.type == "cuda":
            depth_loss_fn = lambda out, gtd: depth_l1_loss(out["depth_acc"], out["alpha"], gtd)
        else:
            depth_loss_fn = lambda out, gtd: cogs_depth_l1(out["depth"], gtd)

    fused_stats = cfg.fused_activations and device.type == "cuda"
    if fused_stats:
        from gs_fused import densify_stats_
    n = n0 = model.num_points
    xys_grad_norm = torch.zeros(n, device=device)
    vis_counts = torch.zeros(n, device=device, dtype=torch.int32)
    max_2dsize = torch.zeros(n, device=device)
    stats_first = cfg.densify  # the reference's `xys_grad_norm is None` (vanilla_gs.py:354)
    rcfg = cfg.refine
    if cfg.densify:
        if rcfg is None:
            from gs_fused import RefineConfig

            rcfg = RefineConfig()
    history = []  # (step, N) after every refinement that changed the model
    max_dim = max(cfg.width, cfg.height)

    depth_err = []  # co-gs: mean |rendered depth - gt depth| over the measured pixels, at every evaluate()

    def evaluate():
        with torch.no_grad():
            idx = np.linspace(0, cfg.num_views - 1, cfg.eval_views).astype(int)
            if not cogs:
                return float(np.mean([psnr(model.render(cams[i], bg, cfg.sh_degree)["rgb"], gt[i]) for i in idx]))
            ps, de = [], []
            for i in idx:
                o = model.render(cams[i], bg, cfg.sh_degree, render_depth=True, fused_depth=device.type == "cuda")
                ps.append(psnr(o["rgb"], gt[i]))
                m = gt_depth[i] > 0
                de.append(float((o["depth"][..., 0] - gt_depth[i]).abs()[m].mean()) if bool(m.any()) else 0.0)
            depth_err.append(float(np.mean(de)))
            return float(np.mean(ps))

    start_step = 0
    if cfg.resume_from:
        from .checkpoint import load_checkpoint

        start_step = load_checkpoint(cfg.resume_from, model, optims, sharded=sharded)  # resizes the model to the saved N
        n = model.num_points
        xys_grad_norm = torch.zeros(n, device=device)
        vis_counts = torch.zeros(n, device=device, dtype=torch.int32)
        max_2dsize = torch.zeros(n, device=device)
    psnr0 = evaluate()
    losses = []
    # gradient exchange: started per parameter from autograd hooks (overlaps the rest of the
    # backward), SH bands above the warm-up degree left out
    exchange = GradientExchange({k: model.gauss[k] for k in PARAM_NAMES}, average=True, force=cfg.force_exchange)
    # a replayed HIP graph fires no hooks, and a hook during capture would put a collective INTO the
    # graph (and reduce every gradient twice): under use_graph the exchange is started after the replay
    exchange.use_hooks = not cfg.use_graph
    if sharded is not None:
        exchange.enabled = False  # the gradients travel by reduce-scatter inside ShardedAdam.step()
    exchange.attach()
    exchanged_bytes = []
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if dp:
        dist.barrier()
    # the ~170 k objects that torch and the setup above leave tracked never die during training: taken out of the
    # cyclic collector's generations for the duration of the loop, so that its full passes (~30 ms each over those
    # objects) do not recur -- the loop's own garbage is still collected
    import gc

    gc.collect()
    gc.freeze()
    if device.type == "cuda":
        torch.cuda.reset_peak_memory_stats(device)
    t0 = time.perf_counter()
    use_fused = (cfg.fused_render or cfg.use_graph) and device.type == "cuda" and cfg.split_sh and cfg.fused_loss \
        and cfg.sh_degree in (0, 1, 2, 3)
    # (through the one native call per view as well, but not under graph replay: no hooks there)
    sh_views = dp and cfg.sh_exchange == "views" and exchange.enabled and exchange.use_hooks
    if sh_views and device.type != "cuda":
        exchange.sh_views_backward = _sh_views_backward_autograd()  # no native kernel here: autograd of the SH op
    fstats = caps = vgraph = vkey = None
    generation = 0
    overflow_views = 0
    if use_fused:
        from gs_fused import DensifyStats, ListCapacity, ViewSpec, render_gaussians
        from gs_fused.render import ViewGraph

        fstats = DensifyStats(model.num_points, device, max_dim)
        caps = ListCapacity()
        graph_loss = lambda out, targets: loss_fn(out["rgb"], targets[0])
        with torch.no_grad():  # size the lists once, synchronously (vanilla: every view, utils.py:124)
            while True:
                g_ = model.gauss
                probe = render_gaussians(g_["means"], g_["scales"], g_["quats"], g_["opacities"], g_["features_dc"],
                                         g_["features_rest"], cams[0].viewmat, cams[0].projmat, cams[0].campos, bg,
                                         ViewSpec(cfg.height, cfg.width, cams[0].fx, cams[0].fy, cams[0].cx,
                                                  cams[0].cy, cfg.sh_degree), caps.capacity)
                need = int(probe["count"].item())
                if need <= caps.capacity:
                    caps.capacity = max(caps.capacity, ((int(1.5 * need) + 65536 + (1 << 20) - 1) >> 20) << 20)
                    break
                caps.capacity = ((int(1.5 * need) + (1 << 20)) >> 20) << 20
    def zero_grads():
        for p_ in model.param_list():
            p_.grad = None

    phase_marks = []
    first_pending, first_vis = True, None
    rebuilds0 = _list_rebuilds()
    target_cache = {}  # (view, downscale factor) -> (resized alpha * rgb [h,w,3], resized 1 - alpha [h,w,1])
    bg_static = bg.clone() if (random_bg and cfg.use_graph) else None  # a replayed graph reads its background here
    try:  # (an exception in the loop must not leave the collector frozen for the rest of the process)
        for step in range(start_step, cfg.iters):
            ph = None
            v = view_for_rank(step, rank, world, cfg.num_views)
            deg = min(step // cfg.sh_degree_interval, cfg.sh_degree)
            exchange.active_rows["features_rest"] = (deg + 1) ** 2 - 1
            # this step's resolution (vanilla_gs.py:646-657, 719-720), background (:688-690) and ground truth
            # (:859-868 downscaled every step, :870-881 composited over the step's background)
            d = downscale_factor(step, cfg.num_downscales, cfg.resolution_schedule)
            cam = cams_by_d[d][v]
            max_dim = max(cam.width, cam.height)  # `max(self.last_size)` of after_train / refinement_after
            if random_bg:
                bg_step = torch.rand(3, device=device, generator=bg_gen)
                if cfg.fused_target:
                    planes = target_cache.get((v, d))
                    if planes is None:
                        small = downscale_image(gt_rgba[v], d)
                        a_ = small[..., 3:4]
                        planes = target_cache[(v, d)] = ((a_ * small[..., :3]).contiguous(), (1 - a_).contiguous())
                    target = torch.addcmul(planes[0], planes[1], bg_step)
                else:
                    target = composite_with_background(downscale_image(gt_rgba[v], d), bg_step)
                if bg_static is not None:
                    bg_static.copy_(bg_step)
                    bg_step = bg_static
            else:
                bg_step, target = bg, downscale_image(gt[v], d)
            if use_fused:
                spec = ViewSpec(cam.height, cam.width, cam.fx, cam.fy, cam.cx, cam.cy, deg)
                fstats.enabled = not (cfg.densify and step >= rcfg.stop_split_at)
                fstats.max_dim = max_dim
                g_ = model.gauss
                if cfg.use_graph:
                    # `generation` counts the refinements that swapped parameter tensors: N can come out
                    # unchanged (k culled, k duplicated) while every tensor the graph points at is gone
                    key = (spec, model.num_points, caps.capacity, fstats.enabled, generation)
                    if vkey != key:  # new SH degree, N changed by refinement, or larger lists: capture again
                        vgraph = ViewGraph({k: g_[k] for k in PARAM_NAMES}, spec, caps.capacity, graph_loss, bg_step,
                                           [(cam.height, cam.width, 3)], stats=fstats)
                        vgraph.capture(cam.viewmat, cam.projmat, cam.campos, (target,))
                        vkey = key
                    else:
                        # the count of the previous replay is in pinned memory by now
                        if not vgraph.fits():
                            overflow_views += 1
                            caps.capacity = ((int(1.5 * int(vgraph.count_host[0])) + (1 << 20)) >> 20) << 20
                    loss, out = vgraph.replay(cam.viewmat, cam.projmat, cam.campos, (target,))
                    if dp:
                        exchange.start_all()
                else:
                    zero_grads()
                    slot = caps.slot(device)
                    used = caps.capacity
                    collector = exchange.begin_sh_views(("features_dc", "features_rest"),
                                                        (g_["features_dc"], g_["features_rest"]), g_["means"], cam.campos,
                                                        cfg.sh_degree, deg) if sh_views else None
                    out = render_gaussians(g_["means"], g_["scales"], g_["quats"], g_["opacities"], g_["features_dc"],
                                           g_["features_rest"], cam.viewmat, cam.projmat, cam.campos, bg_step, spec, used,
                                           count_out=slot, stats=fstats, sh_collector=collector)
                    loss = loss_fn(out["rgb"], target)
                    loss.backward()
                    caps.submitted(slot, used, device)
                    if caps.overflowed():
                        overflow_views += 1
            else:
                zero_grads()
                ph = _phase_marks(5) if (cfg.phase_every and step % cfg.phase_every == 0 and device.type == "cuda") else None
                if ph:
                    ph[0].record()
                depth_on = cogs and cfg.use_depth_loss and step > cfg.depth_loss_start_iteration
                out = model.render(cam, bg_step, deg, retain_xys_grad=True, clamp_rgb=not fused_clamp,
                                   sh_exchange=exchange if sh_views else None, caller_syncs=cfg.caller_syncs,
                                   render_depth=cogs, fused_depth=cogs and cfg.fused_depth and device.type == "cuda",
                                   normalise_depth=not (cogs and cfg.fused_depth and cfg.fused_loss
                                                        and device.type == "cuda" and not cfg.use_est_depth))
                rgb = out["rgb"]
                if ph:
                    ph[1].record()
                loss = loss_fn(rgb, target)
                if cogs and cfg.use_scale_regularization and step % 10 == 0:  # depth_gs.py:450-460
                    loss = loss + cogs_losses.scale_regularisation(model.gauss["scales"], cfg.max_gauss_ratio)
                if cogs and cfg.use_sparse_loss and step % 100 == 0:  # :462-467 (the raw opacity parameter, as written)
                    loss = loss + cogs_losses.sparse_loss(model.gauss["opacities"], cfg.sparse_lambda)
                if depth_on and cfg.use_est_depth:
                    # the monocular-depth branch (:477-531); the synthetic ground truth is metric: scale 1, shift 0
                    terms = cogs_losses.optional_depth_terms(cfg, step, out["depth"], downscale_depth(gt_depth[v], d),
                                                            target, mono_scale_shift=(1.0, 0.0))
                    for term in terms.values():
                        loss = loss + term
                elif depth_on:
                    # `gt_depth = self.get_gt_img(batch["depth"])` (downscaled like the image under a resolution schedule)
                    loss = loss + depth_loss_fn(out, downscale_depth(gt_depth[v], d))
                if ph:
                    ph[2].record()
                loss.backward()
                if ph:
                    ph[3].record()
            # densification statistics (vanilla_gs.py:344-372; not updated past stop_split_at, :347)
            if use_fused or (cfg.densify and step >= rcfg.stop_split_at):
                pass
            elif fused_stats:
                densify_stats_(out["xys"].grad, out["radii"], max_dim, xys_grad_norm, vis_counts, max_2dsize,
                               first=stats_first)
            else:
                with torch.no_grad():
                    visible = out["radii"] > 0
                    g = out["xys"].grad
                    gnorm = torch.zeros_like(xys_grad_norm) if g is None else g.norm(dim=-1)
                    size = out["radii"].float() / max_dim
                    if stats_first:
                        xys_grad_norm, vis_counts = gnorm.clone(), torch.ones_like(vis_counts)
                        max_2dsize = torch.where(visible, size, torch.zeros_like(size))
                    else:
                        xys_grad_norm += torch.where(visible, gnorm, torch.zeros_like(xys_grad_norm))
                        vis_counts += visible.to(torch.int32)
                        max_2dsize = torch.where(visible, torch.maximum(max_2dsize, size), max_2dsize)
            if first_pending and world > 1 and rank > 0 and cfg.densify and step < rcfg.stop_split_at:
                # which Gaussians this rank's FIRST view after a refinement really saw (parallel.single_process_vis_counts)
                first_vis = (out["radii"] > 0).to(torch.int32)
            first_pending = False
            stats_first = False
            if dp and sharded is None:
                b = exchange.finish()
                if not exchanged_bytes or exchanged_bytes[-1][1] != b:
                    exchanged_bytes.append((step, b))
            if cfg.means_lr_schedule:
                # the scheduler steps after the optimizer (trainer.py:479-525): iteration `step` runs at lr(step)
                if sharded is not None:
                    sharded.set_lr("means", means_lr(step, LRS["means"]))
                else:
                    grp = optims["all"].param_groups[0] if "all" in optims else optims["means"].param_groups[0]
                    grp["lr"] = means_lr(step, LRS["means"])
            for o in optims.values():
                o.step()
            if sharded is not None:
                b = sharded.step()
                if not exchanged_bytes or exchanged_bytes[-1][1] != b:
                    exchanged_bytes.append((step, b))
            if ph:
                ph[4].record()
                phase_marks.append((d, ph, bool(cogs and cfg.use_depth_loss and step > cfg.depth_loss_start_iteration)))
            # refinement_after: every refine_every iterations, after the optimizer step
            # (TrainingCallback(update_every_num_iters=refine_every), vanilla_gs.py:610-616)
            if cfg.densify and step % rcfg.refine_every == 0 and step > rcfg.warmup_length:
                branch, reset = _refinement_branch(rcfg, step, cfg.num_views)
                if branch != "none" or reset:
                    if use_fused:
                        xys_grad_norm, vis_counts, max_2dsize = fstats.as_tuple()
                    if dp and branch == "densify":
                        allreduce_densify_stats(xys_grad_norm, vis_counts, max_2dsize, first_visible=first_vis,
                                                force=cfg.force_exchange)
                    old = {k: model.gauss[k] for k in PARAM_NAMES}
                    moments = {}
                    for o in optims.values():
                        moments.update(_adam_moments(o, old))
                    if sharded is not None:
                        moments = sharded.full_moments()  # refinement moves whole rows of both moments
                    new, new_moments, info = _refine(old, moments, (xys_grad_norm, vis_counts, max_2dsize), rcfg, step,
                                                     cfg.num_views, max_dim, seed=cfg.refine_seed + step)
                    if any(new[k] is not old[k] for k in PARAM_NAMES):
                        generation += 1
                        model.replace(new)
                        cur = {k: model.gauss[k] for k in PARAM_NAMES}
                        for o in optims.values():
                            _swap_parameters(o, old, cur, new_moments)
                        if info["n_out"] != info["n_in"]:
                            n = info["n_out"]
                            xys_grad_norm = torch.empty(n, device=device)
                            vis_counts = torch.empty(n, device=device, dtype=torch.int32)
                            max_2dsize = torch.empty(n, device=device)
                            if use_fused:
                                fstats = DensifyStats(n, device, max_dim)
                        history.append((step, model.num_points))
                        exchange.rebind({k: model.gauss[k] for k in PARAM_NAMES})
                    if sharded is not None:
                        # new tensors, or the same ones with an opacity reset (which zeroed moments in the
                        # gathered copies): cut this rank's rows again
                        sharded.bind({k: model.gauss[k] for k in PARAM_NAMES}, new_moments)
                # the statistics restart after every refinement_after past the warm-up (:491-493)
                stats_first = True
                first_pending, first_vis = True, None
                if use_fused:
                    fstats.restart()
            if cfg.save_every and cfg.checkpoint_dir and step > 0 and step % cfg.save_every == 0 and \
                    (rank == 0 or sharded is not None):
                from .checkpoint import save_checkpoint

                # (sharded moments are gathered by a collective: every rank calls, rank 0 writes)
                save_checkpoint(cfg.checkpoint_dir, step, model, optims, sharded=sharded, write=rank == 0)
            if cfg.log_every and step % cfg.log_every == 0:
                losses.append(float(loss.detach()))
        if dp:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
    finally:
        gc.unfreeze()
    psnr1 = evaluate()
    phases = phases_by_res = phases_by_depth = phase_series = None
    if phase_marks:
        if cfg.phase_series:
            phase_series = [[dd] + [round(m[i].elapsed_time(m[i + 1]), 4) for i in range(4)] for dd, m, _ in phase_marks]
        names = ("render", "loss", "backward", "stats_exchange_optimizer")

        def med(marks):
            ms = np.array([[m[i].elapsed_time(m[i + 1]) for i in range(4)] for m in marks])
            out = {k: round(float(np.median(ms[:, i])), 4) for i, k in enumerate(names)}
            out["samples"] = len(marks)
            return out

        phases = med([m for _, m, _ in phase_marks])
        if len(factors) > 1:  # the coarse-to-fine schedule: one set of medians per resolution
            phases_by_res = {f"{cfg.width // d_}x{cfg.height // d_}": med([m for dd, m, _ in phase_marks if dd == d_])
                             for d_ in sorted({dd for dd, _, _ in phase_marks}, reverse=True)}
        if cogs:  # before / after the depth loss joins (the depth pass then has a backward as well)
            phases_by_depth = {("depth_loss_on" if on else "depth_loss_off"): med([m for _, m, o in phase_marks if o == on])
                               for on in sorted({o for _, _, o in phase_marks})}
    if cfg.export_ply and rank == 0:
        from gs_io.ply import write_gaussian_ply

        write_gaussian_ply(cfg.export_ply, {k: model.gauss[k].detach().cpu().numpy() for k in PARAM_NAMES})
    checksum = float(sum(p.detach().double().sum() for p in model.param_list()))
    return {"iters": cfg.iters - start_step, "start_step": start_step, "seconds": elapsed,
            "iters_per_s": (cfg.iters - start_step) / elapsed, "psnr_start": psnr0,
            "psnr_end": psnr1, "losses": losses, "param_checksum": checksum,
            "views_per_s": world * (cfg.iters - start_step) / elapsed, "num_gaussians_start": n0,
            "num_gaussians_end": model.num_points, "refinements": history,
            # (step, bytes) whenever the per-step exchange volume changed: SH warm-up, refinement
            "allreduce_bytes": exchanged_bytes,
            "render": ("hip graph per view" if cfg.use_graph else "one fused op") if use_fused else "separate ops",
            "list_overflow_views": overflow_views + (_list_rebuilds() - rebuilds0), "phase_ms_median": phases,
            "phase_ms_median_by_resolution": phases_by_res, "phase_ms_median_by_depth_loss": phases_by_depth,
            "phase_ms_series": phase_series,
            "depth_segments": _depth_segments_record(device),
            "schedule": {"num_downscales": cfg.num_downscales, "resolution_schedule": cfg.resolution_schedule,
                         "background_color": cfg.background_color, "caller_syncs": cfg.caller_syncs},
            "update": "reduce-scatter + sharded Adam + all-gather" if sharded is not None else
            ("all-reduce (geometry) + all-gathered colour cotangents (SH) + Adam" if sh_views else "all-reduce + Adam"),
            "init": cfg.init, "model": cfg.model,
            "depth": ({"loss_from_step": cfg.depth_loss_start_iteration + 1 if cfg.use_depth_loss else None,
                       "one_compositing_pass": bool(cfg.fused_depth and device.type == "cuda"),
                       "mean_abs_error_start_end": [depth_err[0], depth_err[-1]]} if cogs else None),
            "peak_memory_bytes": (int(torch.cuda.max_memory_allocated(device)) if device.type == "cuda" else None),
            "densify_grad_thresh": (rcfg.densify_grad_thresh if cfg.densify else None)}


def _phase_marks(n):
    return [torch.cuda.Event(enable_timing=True) for _ in range(n)]


def _list_rebuilds() -> int:
    """Views whose device-sized tile lists came out too small and were built again (rasterizer.rasterize counts
    them; 0 where that module's native library is not loaded: CPU stand-ins)."""
    import sys

    mod = sys.modules.get("rasterizer.rasterize")
    return int(mod.counters["list_rebuilds"]) if mod is not None and hasattr(mod, "counters") else 0


# the refinement backend (module-level so that CPU tests can substitute stand-ins)
def _refinement_branch(rcfg, step, num_train_data):
    from gs_fused import refinement_branch

    return refinement_branch(rcfg, step, num_train_data)


def _adam_moments(optimizer, params):
    from gs_fused import adam_moments

    return adam_moments(optimizer, params)


def _swap_parameters(optimizer, old, new, new_moments):
    from gs_fused import swap_parameters

    return swap_parameters(optimizer, old, new, new_moments)


def _refine(params, moments, stats, rcfg, step, num_train_data, max_dim, seed):
    from gs_fused import refine_gaussians

    return refine_gaussians(params, moments, stats, rcfg, step, num_train_data, max_dim, seed=seed)
