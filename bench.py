#!/usr/bin/env python3
"""bench.py -- raster fwd+bwd Mpix/s @1080p, 1 M Gaussians, SH degree 3.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one view, through the public
drop-in API (the three autograd Functions of `rasterizer`):
  forward : project_gaussians -> spherical_harmonics -> clamp(+0.5)
            -> rasterize_gaussians(return_alpha=True)   [scan, key emission,
               radix sort, bin edges, compositing]
  backward: rasterize backward -> SH backward -> project backward
            for fixed cotangents (v_out_img, v_out_alpha)
Inputs (Gaussian parameters, camera, cotangents) are resident in HBM before the
timed region.  Multi-GPU: per-view data parallel -- every rank holds the full
Gaussian set, renders its own camera, and the 59-float/Gaussian gradient is
summed across ranks with one RCCL all-reduce per step (weak scaling).

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gaussian-splatting-toolkit_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from harness import scene as S  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured copy

KERNELS = (
    # rasterizer.cuda attribute, key in the algorithmic-bytes table
    ("project_gaussians_forward", "project_fwd"),
    ("compute_sh_forward", "sh_fwd"),
    ("count_reach", "count_reach"),
    ("depth_order", "depth_order"),
    ("reach_records_depth_order", "depth_order"),  # records + order in one call (the records ride in the sort's first launch)
    ("bin_sorted", "bin_sorted"),
    ("tile_lists_subrange", "bin_sorted"),       # two-round lists (deep scenes): both partitions and the filter
    ("saturation_filter", "bin_sorted"),
    ("rasterize_forward_round", "raster_fwd"),   # ... both compositing rounds
    ("rasterize_backward_two", "raster_bwd"),
    ("rasterize_forward", "raster_fwd"),
    ("rasterize_forward_ex", "raster_fwd"),
    ("rasterize_forward_rgbd", "raster_fwd"),
    ("rasterize_backward", "raster_bwd"),
    ("rasterize_backward_rgbd", "raster_bwd"),
    ("rasterize_backward_det", "raster_bwd"),
    ("compute_sh_backward", "sh_bwd"),
    ("project_gaussians_backward", "project_bwd"),
)


class KernelTimers:
    """HIP-event brackets around every native call.  The native calls enqueue
    on torch's current stream, which is where these events are recorded."""

    def __init__(self):
        import rasterizer.cuda as C

        self.C = C
        self.pairs = {k: [] for _, k in KERNELS}
        self.enabled = False
        self._orig = {}
        for attr, key in KERNELS:
            fn = getattr(C, attr)
            self._orig[attr] = fn
            setattr(C, attr, self._wrap(fn, key))

    def _wrap(self, fn, key):
        def timed(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pairs[key].append((e0, e1))
            return out

        return timed

    def summary_ms(self):
        """mean duration of one call, per stage"""
        return {k: (float(np.mean([a.elapsed_time(b) for a, b in v])) if v else 0.0)
                for k, v in self.pairs.items()}

    def total_ms(self, bracketed_steps):
        """time per STEP, per stage (calls per step x mean duration): what ranks the stages"""
        n = max(1, bracketed_steps)
        return {k: (float(np.sum([a.elapsed_time(b) for a, b in v])) / n if v else 0.0)
                for k, v in self.pairs.items()}


# What one gfx950 SIMD issues depends on the instruction (tools/exp/valubench2.hip, profiles/r05_valubench2.txt; the
# guide's "SIMD-32, two cycles per wave64 instruction" holds for the plain class only):
#   ~2.4 cycles  v_add / v_sub / v_mul / v_fmac / v_fmamk / v_mov / v_and ...            -> 78.6 T lane-ops/s chip-wide
#   ~2.7         v_fma_f32 (VOP3)
#   ~4.2         v_min / v_max, v_cmp, v_cndmask, DPP, v_cvt, anything with an SGPR source, v_pk_* (2 lane-ops per lane)
#   ~8.2         v_exp / v_rcp / v_rsq, v_permlane{16,32}_swap
# and ONE wave alone issues at most one VALU instruction per ~5 cycles.  The peak below is the plain class's.
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9
# mean issue cost of one VALU instruction of the compositing loops as compiled (tools/isa_tally.py with the table above:
# profiles/r05_isa_tally_cycles.txt, 338.2 cycles / 99 and 2591.9 / 793 instructions per loop body)
VALU_CYCLES_PER_INSTRUCTION = {"raster_fwd": 3.42, "raster_bwd": 3.27}
# `value_normalised` = value x (CALIBRATION_REFERENCE / this run's VALU calibration): what the same code would read on a
# box in the reference clock state.  The reference is the calibration of the lease this round's kernels were tuned on
# (profiles/r05_calibration.txt); the step is VALU-issue bound for ~2/3 of its time (compositing), HBM-bound for the rest
CALIBRATION_REFERENCE_VALU_TOPS = 67.0
FIXED_WARMUP_SECONDS = 0.35  # of the workload itself, ahead of the timed steps, whatever --warmup says


def bind_cpus(dev_index, local_rank, mode="auto"):
    """Bind this process (and everything it spawns) to a few cores of its GPU's NUMA node -- what `numactl
    --cpunodebind` / a launcher's per-rank binding does on a multi-socket host.  Round 6 found the "second mode" of the
    host-bound training phases here (profiles/r06_render480_modes.txt): on a 2 x 64-core box an UNBOUND process starts
    on the far socket and has its threads (interpreter, HIP runtime, autograd engine) spread over 256 CPUs -- the 480x270
    render phase with the models' read-backs then reads 0.41-0.61 ms from process to process; bound to eight cores of
    the GPU's socket it reads 0.343-0.356 ms in six processes of six.  The GPU work is the same; only host turn-around
    between read-backs changes.  -> the description for the line's `config.cpu_bind`, or None when nothing was bound
    (`--cpu-bind off`, no sysfs, a single-node host, an affinity mask already narrowed by the launcher)."""
    if mode == "off" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        have = sorted(os.sched_getaffinity(0))
        if len(have) < os.cpu_count():  # somebody (taskset, numactl, the launcher) has already chosen: leave it
            return f"inherited ({len(have)} CPUs)"
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            node = 0
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        # physical cores first (the first range of a node's cpulist; SMT siblings follow), skipping the first eight --
        # where interrupts and housekeeping usually sit -- eight cores per local rank
        first = [c for c in cpus if c in have]
        per = 8
        start = 8 + per * local_rank
        if start + per > len(first):
            start = (per * local_rank) % max(1, len(first) - per + 1)
        chosen = first[start:start + per]
        if len(chosen) < 2:
            return None
        os.sched_setaffinity(0, chosen)
        return f"cpus {chosen[0]}-{chosen[-1]} of NUMA node {node} (GPU {bdf})"
    except Exception as e:  # never cost the line
        return f"not bound ({type(e).__name__})"


def calibration(dev, repeats=5):
    """Two fixed workloads on the bench's stream, right after the timed region (same clock state): a VALU-bound fma
    loop and a 1 GB copy (gsr_calibrate_valu / gsr_calibrate_copy).  Best and median of `repeats`."""
    import ctypes

    from rasterizer.cuda._backend import lib as _native

    L = _native()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    scratch = torch.empty(4096, dtype=torch.float32, device=dev)
    nbytes = 1 << 30
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    valu, copy = [], []
    for i in range(repeats + 1):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        ops = L.gsr_calibrate_valu(32768, 2048, ctypes.c_void_p(scratch.data_ptr()), stream)
        e1.record()
        rc = L.gsr_calibrate_copy(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_size_t(nbytes), stream)
        e2.record()
        torch.cuda.synchronize(dev)
        if ops < 0 or rc != 0:
            return {"error": L.gsr_last_error().decode()}
        if i:  # (the first round is the loop's own warm-up)
            valu.append(ops / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            copy.append(2 * nbytes / (e1.elapsed_time(e2) * 1e-3) / 1e9)
    del a, b
    return {"valu_Tops": round(float(np.median(valu)), 2), "valu_Tops_best": round(max(valu), 2),
            "valu_frac_of_peak": round(float(np.median(valu)) / (VALU_PEAK_LANE_OPS / 1e12), 3),
            "copy_GBps": round(float(np.median(copy)), 1), "copy_GBps_best": round(max(copy), 1),
            "what": "non-packed fp32 fma lane-operations/s of a fixed loop (2048 workgroups x 256 lanes x 8 chains x 32768) "
                    "and read+write GB/s of a 1 GB float4 copy, timed with HIP events on the bench's stream right after "
                    "the timed region; median / best of %d" % repeats,
            "reference_valu_Tops": CALIBRATION_REFERENCE_VALU_TOPS}


def _rnd(v, k):
    return None if v is None else round(v, k)


def parity_vs_oracle(gpu, ref):
    """The timed workload's own results against the CPU oracle's (which `cpu_baseline` has just computed on the same
    scene, camera and cotangents): north_star's bars are 1e-4 abs on the image (pixels whose discrete decisions --
    skip / clamp / termination -- are not within 1e-5 of flipping, `ambig`) and 1e-3 relative on the gradients."""
    ok = ~ref["ambig"]
    img_err = np.abs(gpu["rgb"] - ref["out_img"])
    a_err = np.abs((1.0 - gpu["alpha"]) - ref["final_Ts"])
    visible = ref["radii"] > 0
    rec = {"img_max_abs_stable": float(img_err[ok].max()), "alpha_max_abs_stable": float(a_err[ok].max()),
           "img_max_abs_all_pixels": float(img_err.max()), "ambiguous_px_frac": round(float(ref["ambig"].mean()), 6),
           "projection_bit_identical": bool(np.array_equal(gpu["xys"], ref["xys"]) and np.array_equal(gpu["radii"], ref["radii"])),
           "grad_l2_rel": {}, "grad_max_abs_over_max_ref": {}, "within_1e-3_frac": {}}
    # A colour channel within 1e-5 of the clamp `max(sh + 0.5, 0)` (vanilla_gs.py:822) may legitimately sit on either
    # side of it in two fp32 evaluations, and its whole SH gradient with it (trained models keep dark channels AT the
    # clamp): those Gaussians are left out of the SH gradient's figures, and counted.
    sh_ambig = ref.get("sh_clamp_ambiguous")
    rec["sh_clamp_ambiguous_gaussians"] = int(sh_ambig.sum()) if sh_ambig is not None else None
    for k, g_ref in ref["grads"].items():
        g = gpu["grads"][k]
        if k == "sh_coeffs" and sh_ambig is not None and sh_ambig.any():
            g, g_ref = g.copy(), g_ref.copy()
            g[sh_ambig], g_ref[sh_ambig] = 0.0, 0.0
        err = np.abs(g - g_ref)
        rec["grad_l2_rel"][k] = float(np.linalg.norm((g - g_ref).ravel()) / max(np.linalg.norm(g_ref.ravel()), 1e-30))
        rec["grad_max_abs_over_max_ref"][k] = float(err.max() / max(np.abs(g_ref).max(), 1e-30))
        rel = err / np.maximum(np.abs(g_ref), 1e-4 * np.abs(g_ref).max())
        rec["within_1e-3_frac"][k] = round(float((rel.reshape(len(g_ref), -1).max(axis=1) <= 1e-3)[visible].mean()), 5)
    rec["meets"] = {"image_1e-4_abs": bool(rec["img_max_abs_stable"] < 1e-4 and rec["alpha_max_abs_stable"] < 1e-4),
                    "gradients_1e-3_rel": bool(all(v <= 1e-3 for v in rec["grad_max_abs_over_max_ref"].values())
                                               and all(v <= 1e-4 for v in rec["grad_l2_rel"].values()))}
    rec["what"] = ("this run's GPU image / alpha / gradients of the timed scene vs the CPU oracle on the same inputs: image on "
                   "pixels with no decision within 1e-5 of flipping; gradients as L2-relative error, max error over "
                   "max |ref|, and the share of visible Gaussians within 1e-3 relative (|ref| floored at 1e-4 max|ref|)")
    return rec


def cpu_baseline(sc, cam, bg, v_img, v_alpha, deg, keep=None):
    """The CPU oracle (a port of the reference algorithm, oracle/gsr_oracle.c)
    timed on the host cores over the same workload (one full fwd+bwd)."""
    from oracle import oracle as O

    n = sc["means3d"].shape[0]
    threads = O.num_threads()
    dirs = S.viewdirs_for(sc, cam)
    t0 = time.perf_counter()
    sh = O.compute_sh_forward(n, deg, deg, dirs, sc["sh_coeffs"])
    rgbs = np.maximum(sh + 0.5, 0).astype(np.float32)
    r = O.render_forward(sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat,
                         cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width, 16, rgbs,
                         sc["opacities"], bg, ambig_eps=1e-5 if keep is not None else None)
    t1 = time.perf_counter()
    vxy, vconic, vcol, vop = O.rasterize_backward(
        cam.height, cam.width, 16, r["gaussian_ids_sorted"], r["tile_bins"], r["xys"], r["conics"], rgbs,
        sc["opacities"], bg, r["final_Ts"], r["final_idx"], v_img, v_alpha)
    vsh = O.compute_sh_backward(n, deg, deg, dirs, (vcol * (sh + 0.5 > 0)).astype(np.float32))
    zeros = np.zeros(n, np.float32)
    vproj = O.project_gaussians_backward(n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3],
                                         cam.projmat, cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width,
                                         r["cov3d"], r["radii"], r["conics"], r["compensation"], vxy, zeros, vconic,
                                         zeros)
    t2 = time.perf_counter()
    if keep is not None:  # the arrays the GPU's results for this very workload are checked against (`parity_vs_oracle`)
        keep.update(out_img=r["out_img"], final_Ts=r["final_Ts"], ambig=r["ambig"], radii=r["radii"], xys=r["xys"],
                    sh_clamp_ambiguous=(np.abs(sh + 0.5) < 1e-5).any(axis=1),
                    grads={"xys": vxy, "opacities": vop, "sh_coeffs": vsh, "means3d": vproj[2], "scales": vproj[3],
                           "quats": vproj[4]})
    pix = cam.width * cam.height
    return {
        "value": pix / (t2 - t0) / 1e6,
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": f"1 full fwd+bwd of the same workload ({n} Gaussians, {cam.width}x{cam.height}); "
                  f"fwd {t1 - t0:.2f}s bwd {t2 - t1:.2f}s; host has {os.cpu_count()} logical CPUs",
    }, r["num_intersects"]


# BASELINE config 3 ("gs-train gaussian-splatting on a nerfstudio-style synthetic scene (~1M Gaussians),
# 1xMI355X, 7k iters"; config 4 = the same under per-view data parallelism).  ONE definition: bench.py's
# `train` record, tests/test_gpu_train.py::test_config3_* and tools/train_bench.py --config3 all use it.
def config3(iters=7000, schedule="reference"):
    """schedule="reference": the reference's defaults for resolution and background as well (coarse-to-fine:
    num_downscales 2 / resolution_schedule 2000, vanilla_gs.py:48-53 -- 480x270 until step 2000, 960x540 until 4000,
    then 1920x1080; `background_color="random"`, :50) -- what `gs-train gaussian-splatting` runs.
    schedule="full": 1920x1080 from step 0 over one fixed background (the round-3 record, kept for continuity)."""
    from gs_fused import RefineConfig
    from harness.train import TrainConfig

    ref = schedule == "reference"
    return TrainConfig(
        num_gaussians=CONFIG3["truth_gaussians"], width=1920, height=1080, num_views=48, iters=iters,
        sh_degree=3, sh_degree_interval=1000,          # vanilla_gs.py:67 (reference default)
        num_downscales=2 if ref else 0, resolution_schedule=2000, background_color="random" if ref else "fixed",
        densify=True, refine=RefineConfig(),           # every reference default, densify_grad_thresh = 2e-4 included
        init="sfm", init_gaussians=CONFIG3["seed_points"],  # populate_modules from a sparse point cloud
        means_lr_schedule=True,                        # method_configs.py:98-104
        scene="objects", scene_scale=CONFIG3["truth_scale"], tex_cell=CONFIG3["tex_cell"],
        scene_objects=CONFIG3["objects"], scene_extent=CONFIG3["extent"], cam_radius=CONFIG3["cam_radius"],
        phase_every=50, log_every=500)


# the hidden scene: 140 textured spheres (radius 0.18-0.45) in a ball of radius 2.5, tiled by 3 M flat
# Gaussians; 48 cameras on an orbit of radius 5 (the near half overflows the frame).  The model starts
# from 200 k noisy surface points.  Chosen by tools/exp/exp_seed.sh (round 3): with the reference's
# threshold the model settles near 0.45 M Gaussians whatever the scene (0.15 M when the scene fills 40 %
# of the frame, 0.5 M when it overflows it): the rule stops splitting once a Gaussian's mean screen-space
# gradient x max(W, H) / 2 falls below 2e-4, i.e. at a roughly fixed number of Gaussians per covered pixel.
CONFIG3 = {"truth_gaussians": 3_000_000, "seed_points": 200_000, "truth_scale": (0.003, 0.008), "tex_cell": 0.03,
           "objects": (140, 0.18, 0.45), "extent": 2.5, "cam_radius": 5.0}


def fixed_1m(iters=400):
    """The throughput figure at the size BASELINE names: 1 M Gaussians at 1080p, N FIXED (no refinement),
    the toolkit's full iteration.  Config 3 above follows the reference's rule, which keeps the model
    below 1 M on a 48-view scene; this is the rate at 1 M."""
    from harness.train import TrainConfig

    return TrainConfig(num_gaussians=1_000_000, width=1920, height=1080, num_views=16, iters=iters,
                       sh_degree=3, sh_degree_interval=max(1, iters // 4), phase_every=20)


def refined_1m(iters=1500):
    """1 M Gaussians at 1080p WITH the reference's refinement running (BASELINE config 3's "~1M Gaussians", config 5's
    "densify/prune active"): the model starts as a perturbed copy of a 1 M-Gaussian scene, densify / cull / opacity
    reset follow the reference's schedule and thresholds (warm-up 500, every 100), N moves as the rule decides."""
    from gs_fused import RefineConfig
    from harness.train import TrainConfig

    return TrainConfig(num_gaussians=1_000_000, width=1920, height=1080, num_views=16, iters=iters, sh_degree=3,
                       sh_degree_interval=max(1, iters // 4), densify=True, refine=RefineConfig(), phase_every=20,
                       means_lr_schedule=True)


def cogs_3m_4k(iters=1200):
    """BASELINE config 5 as it reads: "co-gs with depth supervision (render_depth branch), 3M Gaussians, 4K render,
    densify/prune active, 1xMI355X" -- a co-gs TRAINING run (harness.train `model="co-gs"`: DepthGSModel's loop,
    depth_gs.py): 3 M Gaussians (config 3's hidden scene, the model a perturbed copy of it), 3840x2160 from step 0
    (num_downscales 0, :51), random background (:49), depth rasterised on the training path (:99, :345-363), main
    loss 0.8 L1 without the SSIM term (:447-448 as written), depth L1 added unweighted (:532-538), refinement with
    every reference threshold and its schedule (warm-up 500, every 100).  Compressed for a bounded leg, and said so
    in the record: the depth loss joins at step 401 instead of 6 001 and an SH band is switched on every 100
    iterations instead of every 1 000, so that the leg's second half runs the steady-state iteration (SH degree 3,
    depth loss on, refinement firing)."""
    from gs_fused import RefineConfig
    from harness.train import TrainConfig

    return TrainConfig(
        model="co-gs", num_gaussians=CONFIG3["truth_gaussians"], width=3840, height=2160, num_views=12, iters=iters,
        sh_degree=3, sh_degree_interval=100, num_downscales=0, background_color="random",
        densify=True, refine=RefineConfig(), init="perturbed", means_lr_schedule=True,
        depth_loss_start_iteration=400, eval_views=2,
        scene="objects", scene_scale=CONFIG3["truth_scale"], tex_cell=CONFIG3["tex_cell"],
        scene_objects=CONFIG3["objects"], scene_extent=CONFIG3["extent"], cam_radius=CONFIG3["cam_radius"],
        phase_every=20, log_every=100)


def unchanged_caller(cfg):
    """`cfg` with the caller as the toolkit has it: torch ops for the activations, `torch.cat` of the SH features,
    the torch-op L1 + SSIM, one `torch.optim.Adam` per group, `after_train` in torch ops -- and the host read-backs
    of `get_outputs` (intrinsics, `radii.sum() == 0`, `(num_tiles_hit > 0).any()`).  Only the three rasterizer ops
    are this package's."""
    cfg.fused_loss = cfg.fused_adam = cfg.split_sh = cfg.fused_activations = cfg.fused_target = False
    cfg.caller_syncs = "camera"
    return cfg


def train_only(args):
    """`bench.py --train-only`: BASELINE config 3 (N = 1) / config 4 (N > 1) through harness.train.train;
    rank 0 prints the record as one JSON line.  Run as a subprocess of the main bench so that a
    failure here can never cost the raster line."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev_index = local_rank % torch.cuda.device_count()
    cpu_bind = bind_cpus(dev_index, local_rank, args.cpu_bind)  # (a spawned N = 1 leg inherits the raster bench's binding)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = args.backend
    if backend == "auto":
        backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
    from harness.train import train

    cfg = config3(args.train_iters)
    def _small(cfg):  # tests only: the same code path on a scene that trains in seconds
        from gs_fused import RefineConfig

        cfg.num_gaussians, cfg.init_gaussians, cfg.width, cfg.height, cfg.num_views = 40_000, 8_000, 320, 180, 8
        cfg.scene_objects, cfg.scene_scale, cfg.sh_degree_interval, cfg.phase_every = (12, 0.3, 0.6), (0.01, 0.03), 40, 10
        cfg.refine = RefineConfig(warmup_length=30, refine_every=20, reset_alpha_every=6, stop_screen_size_at=200)
        cfg.log_every = 10
        cfg.resolution_schedule = max(1, cfg.iters // 4)  # both resolution switches inside the short run

    if args.train_small:
        _small(cfg)
    cfg.export_ply = args.train_export_ply
    cfg.phase_series = True
    res = train(cfg, dev, rank, world)
    full_run = args.train_iters >= 1000 and not args.train_small
    res1m = train(fixed_1m(), dev, rank, world) if full_run else None
    extra = {}
    if full_run or args.train_small:
        # the same configuration with the host blocked where the unchanged models block it (vanilla_gs.py:784,811)
        cfg_s = config3(args.train_iters)
        if args.train_small:
            _small(cfg_s)
        cfg_s.caller_syncs = True
        cfg_s.phase_series = True
        extra["caller_syncs"] = train(cfg_s, dev, rank, world)
    if full_run:
        extra["full_res"] = train(config3(args.train_iters, schedule="full"), dev, rank, world)
        # (a rate, not a quality figure: 2 100 iterations with the three resolution phases in the reference's 2:2:3
        # proportion -- at ~160 it/s the full 7 000 took 44 s of the driver's run)
        cfg_u = unchanged_caller(config3(min(args.train_iters, 2100)))
        cfg_u.resolution_schedule = max(1, cfg_u.iters * 2 // 7)
        extra["unchanged_caller"] = train(cfg_u, dev, rank, world)
        extra["refined_1m"] = train(refined_1m(), dev, rank, world)
    if (full_run or args.train_small) and not args.no_cogs:
        # BASELINE config 5: the co-gs training loop at 3 M Gaussians / 4K (tests: the same code on a tiny scene)
        cfg_c = cogs_3m_4k(args.cogs_iters)
        if args.train_small:
            _small(cfg_c)
            cfg_c.iters, cfg_c.depth_loss_start_iteration, cfg_c.init_gaussians = 120, 40, None
        try:
            extra["cogs_3m_4k"] = (cfg_c, train(cfg_c, dev, rank, world))
        except Exception as e:  # this leg must not cost the config-3 record
            extra["cogs_error"] = repr(e)
    res_one = None
    if full_run:
        # the same configuration through gs_fused.render_gaussians: one autograd node and one native call per view
        # instead of the models' op-by-op sequence (optional API; the headline figure above is the public ops)
        cfg_one = config3(args.train_iters)
        cfg_one.fused_render = True
        res_one = train(cfg_one, dev, rank, world)
    if world > 1:
        cs = torch.tensor([res["param_checksum"]], dtype=torch.float64, device=dev)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res["replicas_identical"] = bool(lo.item() == hi.item())
    if rank == 0:
        losses = res.pop("losses", None) or [None]
        hist = res.pop("refinements")
        rec = {
            "metric": "train iters/s (BASELINE config 3" + (" / 4" if world > 1 else "") + ")",
            "iters_per_s": round(res["iters_per_s"], 1), "views_per_s": round(res["views_per_s"], 1),
            "n_gpus": world, "iters": res["iters"], "seconds": round(res["seconds"], 2),
            "resolution": f"{cfg.width}x{cfg.height}",
            "gaussians": {"start": res["num_gaussians_start"], "end": res["num_gaussians_end"],
                          "max": max([n for _, n in hist] + [res["num_gaussians_start"]]),
                          "history_step_N": hist[:: max(1, len(hist) // 24)]},
            "refinements": len(hist),
            "psnr": {"start": round(res["psnr_start"], 2), "end": round(res["psnr_end"], 2)},
            "loss_first_last": [losses[0], losses[-1]],
            "phase_ms_median": res["phase_ms_median"],
            "phase_ms_median_by_resolution": res["phase_ms_median_by_resolution"],
            "schedule": res["schedule"], "depth_segments": res.get("depth_segments"),
            "list_overflow_views": res["list_overflow_views"],
            "render": res["render"],
            "scene": (f"hidden truth: {cfg.num_gaussians} flat textured Gaussians on {cfg.scene_objects[0]} spheres in a ball of "
                      f"radius {cfg.scene_extent} (harness.train.blob_scene 'objects'), {cfg.num_views} views from an orbit of "
                      f"radius {cfg.cam_radius} rendered by this rasterizer"),
            "why_not_1M": ("every reference default incl. densify_grad_thresh 2e-4, the coarse-to-fine resolution schedule and "
                           "the random background: the rule stops splitting at a roughly fixed number of Gaussians per "
                           "covered pixel (tools/exp/exp_seed.sh, DESIGN 4.7); the rate at 1 M Gaussians is in `fixed_1m` "
                           "(N fixed) and `refined_1m` (refinement active)"),
            "model_start": (f"{cfg.init}: {cfg.init_gaussians} noisy surface points with 8-bit colours; scales = 3-NN distance, "
                            "random quats, opacity 0.1 (vanilla_gs.py:128-174)"),
            "refinement": {"densify_grad_thresh": res["densify_grad_thresh"], "schedule": "reference defaults "
                           "(warm-up 500, every 100, opacity reset every 3000, screen-size rules until 4000)"},
            "ground_truth": ("RGBA (straight colour + alpha, as a blender-style dataset stores it), downscaled every step "
                             "(vanilla_gs.py:659-670) and composited over the step's random background (:870-881)"
                             if cfg.background_color == "random" else "RGB over the fixed background"),
            "allreduce_bytes_step_bytes": res["allreduce_bytes"] or None,
            "replicas_identical": res.get("replicas_identical"),
            "parallelism": (f"dp{world} per-view, {backend}" if world > 1 else "single"),
            "cpu_bind": cpu_bind,
            "update": res["update"],
        }
        brief = lambda r: {"iters_per_s": round(r["iters_per_s"], 1), "gaussians_end": r["num_gaussians_end"],  # noqa: E731
                           "gaussians_max": max([n_ for _, n_ in r["refinements"]] + [r["num_gaussians_start"]]),
                           "psnr": [round(r["psnr_start"], 2), round(r["psnr_end"], 2)],
                           "list_overflow_views": r["list_overflow_views"], "phase_ms_median": r["phase_ms_median"],
                           "phase_ms_median_by_resolution": r["phase_ms_median_by_resolution"]}
        if "caller_syncs" in extra:
            rec["iters_per_s_with_caller_syncs"] = round(extra["caller_syncs"]["iters_per_s"], 1)
            # The render phase at the schedule's LOWEST resolution had two modes from process to process in round 5
            # (0.37 / 0.53 ms): the share of this leg's samples there that exceed 1.3 x the median
            # of the same phase in the leg WITHOUT read-backs (same process, same kernels) says which one this run saw.
            slow_share = None
            try:
                a_, b_ = res.get("phase_ms_series") or [], extra["caller_syncs"].get("phase_ms_series") or []
                dmax = max(r_[0] for r_ in b_)
                ref = float(np.median([r_[1] for r_ in a_ if r_[0] == dmax]))
                mine = np.array([r_[1] for r_ in b_ if r_[0] == dmax])
                slow_share = round(float((mine > 1.3 * ref).mean()), 3)
            except Exception:
                pass
            rec["with_caller_syncs"] = dict(brief(extra["caller_syncs"]), render_slow_share_lowest_resolution=slow_share, what=(
                "the same run with the host blocked where the unchanged models block it: `if (self.radii).sum() == 0` "
                "and `assert (num_tiles_hit > 0).any()` (vanilla_gs.py:784,811)"))
        if "full_res" in extra:
            rec["full_resolution_from_step_0"] = dict(brief(extra["full_res"]), what=(
                "num_downscales 0, fixed background: the round-3 record's configuration"))
        if "unchanged_caller" in extra:
            rec["iters_per_s_unchanged_caller"] = round(extra["unchanged_caller"]["iters_per_s"], 1)
            rec["unchanged_caller"] = dict(brief(extra["unchanged_caller"]), what=(
                "only the three rasterizer ops are this package's: torch-op activations / torch.cat / torch-op L1+SSIM / "
                "six torch.optim.Adam / torch-op after_train, and all of get_outputs' host read-backs (intrinsics "
                ".item(), radii.sum() == 0, (num_tiles_hit > 0).any())"))
        if "cogs_error" in extra:
            rec["cogs_3m_4k"] = {"error": extra["cogs_error"]}
        if "cogs_3m_4k" in extra:
            c_, r_ = extra["cogs_3m_4k"]
            rec["cogs_3m_4k"] = dict(
                brief(r_), iters=r_["iters"], resolution=f"{c_.width}x{c_.height}", views=c_.num_views,
                gaussians_start=r_["num_gaussians_start"],
                history_step_N=r_["refinements"][:: max(1, len(r_["refinements"]) // 12)],
                refinements=len(r_["refinements"]), seconds=round(r_["seconds"], 2),
                peak_memory_GB=(round(r_["peak_memory_bytes"] / 1e9, 2) if r_.get("peak_memory_bytes") else None),
                depth=r_["depth"], phase_ms_median_by_depth_loss=r_["phase_ms_median_by_depth_loss"],
                loss_first_last=[(r_["losses"] or [None])[0], (r_["losses"] or [None])[-1]],
                what=("BASELINE config 5 as a TRAINING run: harness.train model='co-gs' (DepthGSModel's loop, depth_gs.py) "
                      "-- depth rasterised on the training path in the same compositing pass as the colour, main loss "
                      "(1 - ssim_lambda) L1 with the SSIM term dropped as the source drops it (:447-448), depth L1 on "
                      "gt > 0 added unweighted (:532-538), full resolution from step 0, random background, refinement "
                      "with the reference's thresholds and schedule (warm-up 500, every 100)"),
                compressed=("bounded leg: depth loss from step %d (reference: 6 001), one SH band per %d iterations "
                            "(reference: 1 000)" % (c_.depth_loss_start_iteration + 1, c_.sh_degree_interval)))
        if "refined_1m" in extra:
            r_ = extra["refined_1m"]
            rec["refined_1m"] = dict(brief(r_), iters=r_["iters"], gaussians_start=r_["num_gaussians_start"],
                                     history_step_N=r_["refinements"][:: max(1, len(r_["refinements"]) // 12)],
                                     what="1 M Gaussians at 1920x1080 with the reference's refinement schedule and "
                                          "thresholds active (densify / cull / opacity reset), full training iteration")
        if res_one is not None:
            rec["one_op_path"] = {"what": "config 3 through gs_fused.render_gaussians (one native call per view, statistics "
                                          "from its backward)", "iters_per_s": round(res_one["iters_per_s"], 1),
                                  "gaussians_end": res_one["num_gaussians_end"],
                                  "psnr": [round(res_one["psnr_start"], 2), round(res_one["psnr_end"], 2)],
                                  "list_overflow_views": res_one["list_overflow_views"]}
        if res1m is not None:
            rec["fixed_1m"] = {"workload": "1 M Gaussians at 1920x1080, N fixed (no refinement), full training iteration",
                               "iters_per_s": round(res1m["iters_per_s"], 1), "views_per_s": round(res1m["views_per_s"], 1),
                               "iters": res1m["iters"], "psnr": [round(res1m["psnr_start"], 2), round(res1m["psnr_end"], 2)],
                               "phase_ms_median": res1m["phase_ms_median"],
                               "allreduce_bytes_step_bytes": res1m["allreduce_bytes"] or None}
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


def train_record(args, world):
    """Run `--train-only` in a fresh process (own process group, own spawn) -> dict for the line."""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE") and not k.startswith("TORCHELASTIC")}
    import tempfile

    ply = None
    if world == 1 and not args.no_trained_raster:
        ply = os.path.join(tempfile.gettempdir(), f"gsr_config3_trained_{os.getpid()}.ply")
    cmd = [sys.executable, os.path.abspath(__file__), "--train-only", "--gpus", str(world), "--train-iters",
           str(args.train_iters), "--backend", args.backend, "--cogs-iters", str(args.cogs_iters)] \
        + (["--train-small"] if args.train_small else []) + (["--no-cogs"] if args.no_cogs else []) \
        + ["--cpu-bind", args.cpu_bind] + (["--cpu-bind-reset"] if world > 1 else []) \
        + (["--train-export-ply", ply] if ply else [])
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=args.train_timeout)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"error": f"rc {out.returncode}", "stderr_tail": out.stderr[-800:]}
        rec = json.loads(lines[-1])
        rec["wall_s_including_setup"] = round(time.perf_counter() - t0, 1)
        if ply and os.path.exists(ply):
            try:
                rec["trained_raster"] = trained_raster(args, ply, env)
            finally:
                os.remove(ply)
        return rec
    except subprocess.TimeoutExpired:
        return {"error": f"timeout after {args.train_timeout} s"}
    except Exception as e:  # the raster line must survive anything that goes wrong here
        return {"error": repr(e)}


def trained_raster(args, ply, env):
    """The raster bench (this script, same timed region, counter passes included) on the model the config-3 leg has
    JUST trained, from one of its training views: the distribution `gs-train` really produces (larger, opaque,
    mutually occluding splats) under the driver's clock, next to the random cloud of the headline."""
    import subprocess

    small = ["--width", "320", "--height", "180", "--ply-views", "8"] if args.train_small else []
    # (the CPU oracle runs on this model too -- a few seconds on the host's cores: `parity_vs_oracle` of the one
    #  realistic list distribution the line has, with the job order, tail splitting and the measured split ratio
    #  active; VERDICT r5 item 2.  The one-thread baseline is skipped.)
    cmd = [sys.executable, os.path.abspath(__file__), "--scene", f"ply:{ply}", "--steps", "50", "--warmup", "10",
           "--train-iters", "0", "--no-cpu-one-thread", "--no-synced-regions"] + small + (["--no-pmc"] if args.no_pmc else [])
    try:
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"error": f"rc {out.returncode}", "stderr_tail": out.stderr[-600:]}
        r = json.loads(lines[-1])
        k, rf = r["kernels"], r["roofline"]
        return {"ms": r["ms_per_step"], "ms_median": r["ms_per_step_median"], "Mpix_per_s": r["value"],
                "Mpix_per_s_normalised": r.get("value_normalised"),
                "gaussians": r["config"]["gaussians"], "visible": r["config"]["visible"],
                "intersections": r["config"]["intersections"], "list_entries": r["config"]["list_entries"],
                "tile_list_length": r["config"]["tile_list_length"],
                "raster_fwd_ms": k.get("raster_fwd", {}).get("ms"), "raster_bwd_ms": k.get("raster_bwd", {}).get("ms"),
                "kernels_ms": {n_: v_["ms"] for n_, v_ in k.items()},
                "staged_list_entries": rf.get("staged_list_entries"),
                "dominant": rf["kernel"], "valu_busy": rf.get("valu_busy"), "valu": rf.get("valu"),
                "traffic": rf.get("traffic"), "algorithmic_bytes": rf.get("algorithmic_bytes"), "frac_hbm": rf.get("frac"),
                "calibration": r["config"].get("calibration"),
                "parity_vs_oracle": r.get("parity_vs_oracle"),
                "cpu_baseline": r.get("cpu_baseline"),
                "what": "bench.py --scene ply:<the model this run's config-3 leg ended with>, view 0 of its training orbit, "
                        "50 timed steps after the fixed warm-up; its results against the CPU oracle on the same model"}
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    except Exception as e:
        return {"error": repr(e)}


def _spawned(rank, world, port, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.argv = argv
    main()


def cpu_baseline_one_thread(sc, cam, bg, v_img, v_alpha, deg, budget_gaussians=60_000):
    """The same port on ONE host thread (SURVEY 8d asks for OMP_NUM_THREADS = all and = 1),
    on a bounded sample: the first `budget_gaussians` Gaussians of the same scene, same camera
    and image size (about 10-30 s of CPU work; the full workload takes minutes on one thread)."""
    from oracle import oracle as O

    n = min(budget_gaussians, sc["means3d"].shape[0])
    sub = {k: np.ascontiguousarray(v[:n]) for k, v in sc.items()}
    before = O.num_threads()
    O.set_threads(1)
    try:
        res, I = cpu_baseline(sub, cam, bg, v_img, v_alpha, deg)
    finally:
        O.set_threads(before)
    res["cores"] = 1
    res["not_comparable_with"] = "cpu_baseline: this is a DIFFERENT, smaller workload (a 60 k-Gaussian subset)"
    res["sample"] = (f"first {n} Gaussians of the workload ({I} reference list entries) at {cam.width}x{cam.height}, one "
                     f"thread; " + res["sample"].split("; ", 1)[1])
    return res


# stage -> substrings of the kernel names it launches (rocprofv3 kernel trace)
STAGE_KERNELS = {
    "project_fwd": ("project_fwd_kernel",), "sh_fwd": ("sh16_fwd_kernel", "sh_fwd_kernel", "sh_split_fwd_kernel"),
    "count_reach": ("reach_records_kernel", "tile_rows_kernel<false>"),
    "depth_order": ("gsr_sort::", "gsr_bsort::", "depth_keys_kernel"),
    "bin_sorted": ("gsr_p2::", "gsr_ts::", "tile_rows_kernel<true>", "publish_int_kernel", "tile_flag_", "saturation_filter_kernel"),
    "raster_fwd": ("raster_fwd_tile16_kernel", "raster_fwd_generic_kernel", "raster_fwd_seg"),
    "raster_bwd": ("raster_bwd_tile16_kernel", "raster_bwd_generic_kernel", "reduce_partials_kernel", "raster_bwd_seg"),
    "sh_bwd": ("sh16_bwd_kernel", "sh_bwd_kernel", "sh_split_bwd_kernel"), "project_bwd": ("project_bwd_kernel",),
}


def pmc_pass(counters, argv, timeout_s=240):
    """One separate `rocprofv3 --kernel-trace --pmc <counters>` pass over a short run of this
    script (MI355X_MICROARCH.md, HBM section: counters in their own pass, kernel trace only).
    -> {kernel name: {counter: [value per launch, ...]}} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="gsr_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.abspath(__file__), *argv]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=timeout_s, check=True)
        vals = {}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                vals.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(
                    float(row["Counter_Value"]))
        return vals or None
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pmc_steps(vals, counter, fallback):
    """How many steps the counter sub-run executed: one projection launch per step, counted in the pass's own trace (the
    sub-run also executes the fixed-duration warm-up, hundreds of steps: VERDICT r5, weak 7 -- `traffic_per_step` read
    17x high while this was the constant 6)."""
    if not vals:
        return fallback
    n = sum(len(v.get(counter, ())) for k, v in vals.items() if "project_fwd_kernel" in k)
    return n if n > 0 else fallback


def pmc_stage(vals, stage, counter, steps):
    """-> (mean per launch of the stage's LARGEST kernel, sum over all the stage's launches per step)"""
    if not vals:
        return None, None
    per_kernel = {k: v[counter] for k, v in vals.items()
                  if counter in v and any(sub in k for sub in STAGE_KERNELS[stage])}
    if not per_kernel:
        return None, None
    top = max(per_kernel.values(), key=lambda v: float(np.mean(v)))
    return float(np.mean(top)), float(sum(np.sum(v) for v in per_kernel.values())) / max(1, steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    # SURVEY.md 8(d) prescribes scales ~ exp(U(ln .005, ln .05)) AND that the
    # measured I/N at 1080p must fall in [4,12], else "rescale the scale range
    # and say so".  The prescribed range gives I/N = 21.7 (mean radius 29 px),
    # so the default here is the range halved: I/N = 7.7.  `--scale-lo 0.005
    # --scale-hi 0.05` runs the denser variant (numbers for both in DESIGN.md).
    ap.add_argument("--scale-lo", type=float, default=0.0025)
    ap.add_argument("--scale-hi", type=float, default=0.025)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-one-thread", action="store_true",
                    help="skip the one-thread CPU baseline on the 60 k-Gaussian subset (10-30 s); the all-threads baseline "
                         "and `parity_vs_oracle` stay")
    ap.add_argument("--cpu-bind", default="auto", choices=["auto", "off"],
                    help="auto: bind each rank to eight cores of its GPU's NUMA node before anything runs (bench.bind_cpus: "
                         "the host-bound training phases have two modes on a multi-socket host otherwise); off: leave the "
                         "affinity as inherited")
    ap.add_argument("--cpu-bind-reset", action="store_true", help=argparse.SUPPRESS)  # (the N > 1 training leg: its ranks
    # bind themselves and must not all inherit the cores of the rank that spawned the leg)
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 counter passes (HBM traffic, VALU busy) that rank 0 runs after the timed "
                         "region at N=1")
    ap.add_argument("--event-every", type=int, default=10,
                    help="bracket the native calls with HIP events on every k-th timed step (0: never, 1: all)")
    # co-gs / eval pattern (BASELINE config 5): a second rasterisation of the depths with
    # zero background (depth_gs.py:99, vanilla_gs.py:839-855), differentiable, in the step
    ap.add_argument("--render-depth", action="store_true")
    ap.add_argument("--fused-depth", action="store_true",
                    help="with --render-depth: RGB and depth from ONE compositing pass (gs_fused, SURVEY 8f row f4)")
    # "nccl" is RCCL on ROCm.  "gloo" exists so the N>1 code path can be exercised on a
    # single-GPU box (ranks then share cuda:0); it is not a measurement configuration.
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="auto: nccl (= RCCL) when every rank has its own GPU, gloo otherwise")
    ap.add_argument("--sh-degree-to-use", type=int, default=None,
                    help="evaluate only the first bands (the models' SH warm-up, vanilla_gs.py:811-820); the "
                         "gradient exchange then leaves the inactive bands out")
    ap.add_argument("--sh-exchange", default="views", choices=["views", "dense"],
                    help="N > 1: how the SH gradient crosses ranks -- 'views': all-gather of the 12-byte colour cotangents, "
                         "the sum over views formed on every rank; 'dense': all-reduce of the 12 K-byte gradient")
    ap.add_argument("--deterministic", action="store_true",
                    help="compositing backward with a fixed summation order (gsr_rasterize_backward_det) instead of "
                         "float atomics")
    ap.add_argument("--scene", default="uniform",
                    help="uniform (SURVEY 8d's random cloud) | longtail: 10 %% of the tiles hold ~10x the list depth "
                         "(clustered Gaussians) | room | floaters | needles: held-out families (harness/scene.py) | ply:<path>: a TRAINED model in the toolkit's export format "
                         "(gs_io/ply.py; scripts/exporter.py:88-147), seen from --ply-view of an orbit of radius "
                         "--ply-cam-radius (the cameras harness.train trains on)")
    ap.add_argument("--ply-cam-radius", type=float, default=5.0)
    ap.add_argument("--ply-view", type=int, default=0)
    ap.add_argument("--ply-views", type=int, default=48)
    ap.add_argument("--caller-syncs", default="off", choices=["off", "on", "camera"],
                    help="run the MAIN timed region with the unchanged models' host read-backs in the caller (profiling "
                         "runs; the default line reports all three modes anyway)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="with --gpus 1: create the process group anyway (one rank) and run the gradient exchange "
                         "through it -- every collective of the N > 1 path goes through the backend as an identity "
                         "(how the RCCL path is exercised on a single-GPU box)")
    ap.add_argument("--no-synced-regions", action="store_true",
                    help="skip the two extra timed regions with the caller's read-backs (profiling runs)")
    ap.add_argument("--train-iters", type=int, default=7000,
                    help="iterations of the config-3 training record (BASELINE metric, second half); 0: skip it")
    ap.add_argument("--train-timeout", type=int, default=900)
    ap.add_argument("--train-only", action="store_true", help="run only the config-3 training leg and print its record")
    ap.add_argument("--train-export-ply", default=None,
                    help="--train-only: write the model the config-3 run ends with to this PLY (then: --scene ply:<path>)")
    ap.add_argument("--cogs-iters", type=int, default=1200, help="iterations of the co-gs 3 M / 4K training leg (config 5)")
    ap.add_argument("--no-cogs", action="store_true", help="skip the co-gs leg of the training record")
    ap.add_argument("--no-trained-raster", action="store_true",
                    help="skip the raster bench on the model the config-3 leg has just trained")
    ap.add_argument("--train-small", action="store_true", help=argparse.SUPPRESS)  # tests: a scene that trains in seconds
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.cpu_bind_reset and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, range(os.cpu_count()))
        except OSError:
            pass
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly (`python bench.py --gpus N`): spawn one rank per GPU ourselves, as the
        # reference's launcher does (gs_toolkit/scripts/train.py:169, torch.multiprocessing.spawn)
        import socket

        import torch.multiprocessing as mp

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        mp.spawn(_spawned, args=(args.gpus, port, list(sys.argv)), nprocs=args.gpus, join=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.train_only:
        return train_only(args)
    dev_index = local_rank % torch.cuda.device_count()
    cpu_bind = bind_cpus(dev_index, local_rank, args.cpu_bind)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = args.backend
    if backend == "auto":
        backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    args.backend = backend
    dp = world > 1 or args.force_exchange
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    from harness.parallel import GradientExchange
    from harness.pipeline import CameraTensors, render_view

    timers = KernelTimers()
    if args.deterministic:
        from rasterizer import rasterize as _Rd

        _Rd.set_deterministic(True)

    # ---- workload: SURVEY.md 8(d), seed 42; one scene, one camera per rank
    W, H, N, deg = args.width, args.height, args.gaussians, args.sh_degree
    if args.scene.startswith("ply:"):
        # a trained model: raw parameters from the file, activated as get_outputs activates them
        # (vanilla_gs.py:765-830: exp, normalise, sigmoid, cat of the SH features)
        from gs_io.ply import read_gaussian_ply
        from harness.train import orbit_cameras

        raw = read_gaussian_ply(args.scene[4:])
        N = raw["means"].shape[0]
        deg = {0: 0, 3: 1, 8: 2, 15: 3}[raw["features_rest"].shape[1]]
        q = raw["quats"] / np.linalg.norm(raw["quats"], axis=-1, keepdims=True)
        sc = {"means3d": raw["means"], "scales": np.exp(raw["scales"]).astype(np.float32), "quats": q.astype(np.float32),
              "opacities": (1.0 / (1.0 + np.exp(-raw["opacities"].astype(np.float64)))).astype(np.float32),
              "sh_coeffs": np.ascontiguousarray(np.concatenate([raw["features_dc"][:, None, :], raw["features_rest"]], 1))}
        cams = orbit_cameras(args.ply_views, W, H, radius=args.ply_cam_radius)
        cam0 = cams[args.ply_view % len(cams)]
        cam = cams[(args.ply_view + rank * 5) % len(cams)]
    elif args.scene == "ball":
        # the trainer's default scene (harness.train.blob_scene "ball", the fixed_1m leg's): translucent Gaussians
        # filling a ball, seen from the training orbit -- deep lists in the middle of the frame, empty tiles around it
        from harness.train import blob_scene, orbit_cameras

        raw = blob_scene(N, seed=0, sh_degree=deg)
        q = raw["quats"] / np.linalg.norm(raw["quats"], axis=-1, keepdims=True)
        sc = {"means3d": raw["means"], "scales": np.exp(raw["scales"]).astype(np.float32), "quats": q.astype(np.float32),
              "opacities": (1.0 / (1.0 + np.exp(-raw["opacities"].astype(np.float64) + 0.5))).astype(np.float32),
              "sh_coeffs": np.ascontiguousarray(np.concatenate([raw["features_dc"][:, None, :], raw["features_rest"]], 1))}
        cams = orbit_cameras(16, W, H, radius=6.0)
        cam0 = cams[0]
        cam = cams[(rank * 5) % len(cams)]
    elif args.scene in ("uniform", "longtail"):
        cam0 = S.make_camera(W, H)
        sc = S.make_scene(N, cam0, sh_degree=deg, seed=42, scale_lo=args.scale_lo, scale_hi=args.scale_hi,
                          longtail=args.scene == "longtail")
        # rank r looks at the same cloud from a slightly different direction
        cam = cam0 if rank == 0 else S.make_camera(W, H, yaw=0.02 * rank, pitch=0.01 * (rank % 3))
    elif args.scene in S.HELDOUT_KINDS:
        # scene families no dispatch constant was fitted on (harness.scene.make_heldout_scene)
        cam0 = S.make_camera(W, H)
        sc = S.make_heldout_scene(args.scene, N, cam0, sh_degree=deg)
        cam = cam0 if rank == 0 else S.make_camera(W, H, yaw=0.02 * rank, pitch=0.01 * (rank % 3))
    else:
        raise SystemExit(f"unknown --scene {args.scene!r}")
    bg_np = np.array(S.BACKGROUND, np.float32)
    v_img_np, v_alpha_np = S.make_cotangents(cam)

    t = lambda a: torch.from_numpy(a).to(dev)
    params = {k: t(v).requires_grad_(True) for k, v in sc.items()}
    plist = [params[k] for k in ("means3d", "scales", "quats", "opacities", "sh_coeffs")]
    camt = CameraTensors.from_numpy(cam, dev)
    bg, v_img, v_alpha = t(bg_np), t(v_img_np), t(v_alpha_np)

    comm_events = []
    deg_use = deg if args.sh_degree_to_use is None else min(args.sh_degree_to_use, deg)
    # the exchange starts per parameter from autograd hooks, as soon as a gradient exists (the SH
    # block while project_backward still runs), and leaves inactive SH bands out
    exchange = GradientExchange({k: params[k] for k in ("means3d", "scales", "quats", "opacities", "sh_coeffs")},
                                average=True, force=args.force_exchange).attach()
    exchange.active_rows["sh_coeffs"] = (deg_use + 1) ** 2
    # N > 1: the SH gradient is formed on every rank from the ranks' all-gathered 12-byte colour cotangents instead of
    # being all-reduced (GradientExchange `sh_views`; --sh-exchange dense: every gradient all-reduced)
    sh_views = dp and args.sh_exchange == "views" and deg <= 3

    main_mode = {"off": False, "on": True, "camera": "camera"}[args.caller_syncs]

    def step(caller_syncs=main_mode):
        for p in plist:
            p.grad = None
        out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"],
                          params["sh_coeffs"], camt, bg, deg_use, clamp_rgb=False, render_depth=args.render_depth,
                          fused_depth=args.fused_depth, caller_syncs=caller_syncs,
                          sh_exchange=(exchange, ("sh_coeffs",), (params["sh_coeffs"],)) if sh_views else None)
        if args.render_depth:
            torch.autograd.backward([out["rgb"], out["alpha"], out["depth"]],
                                    [v_img, v_alpha[..., None], v_alpha[..., None]])
        else:
            torch.autograd.backward([out["rgb"], out["alpha"]], [v_img, v_alpha[..., None]])
        if dp:
            if timers.enabled:  # what is left of the exchange once the backward has been queued
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                exchange.finish()
                e1.record()
                comm_events.append((e0, e1))
            else:
                exchange.finish()
        return out

    def barrier():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):  # (at least one: the workload's statistics below come from a rendered view)
        out = step()
    num_intersects = int(out["num_tiles_hit"].sum().item())  # the reference's lists (3-sigma boxes)
    from rasterizer import rasterize as _R
    list_entries = int(_R._bin_cache["value"][0])  # what the kernels walk (dead pairs left out)
    n_visible = int((out["radii"] > 0).sum().item())
    _aux = _R._bin_cache["value"][3]
    two_round = None
    if isinstance(_aux, tuple) and _aux and _aux[0] == "two":  # deep scene: lists in two segments
        _th = next(iter(_R._two_hint.values()), {})
        two_round = {"prefix_fraction": round(_th.get("f_used", 0.0), 4), "entries_round1": _th.get("count1"),
                     "entries_round2": _th.get("count2"), "tiles_unfinished_after_round1": _th.get("unfinished")}
    _bins = _R._bin_cache["value"][2]
    _lens = (_bins[:, 1] - _bins[:, 0]).float().cpu().numpy()
    tile_hist = {"p50": int(np.percentile(_lens, 50)), "p90": int(np.percentile(_lens, 90)),
                 "p99": int(np.percentile(_lens, 99)), "max": int(_lens.max()), "mean": round(float(_lens.mean()), 1)}

    # list entries the compositing kernels really stage (tiles stop once saturated): one
    # untimed step with the library's measurement hook on
    import ctypes

    from rasterizer.cuda._backend import lib as _native

    staged = torch.zeros(2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _native().gsr_debug_count_staged(ctypes.c_void_p(staged.data_ptr()))
    step()
    torch.cuda.synchronize()
    _native().gsr_debug_count_staged(None)
    raster_launches = 2 if (args.render_depth and not args.fused_depth) else 1
    staged_fwd, staged_bwd = (int(v) // raster_launches for v in staged.tolist())

    # A full pass of Python's cyclic collector over the ~170 k objects torch leaves tracked takes ~30 ms: inside a
    # 100-ms timed region that is 0.3 ms per step of pure artefact (tools/exp/gc_frames.py: it decided whether the
    # forward-only bench read 0.50 or 0.67 ms).  Collect NOW, so that none falls due inside the region; the collector
    # stays on.
    import gc

    gc.collect()
    # (one bracketed step now: the per-kernel HIP events are created on first use, and the first timed step is a
    #  bracketed one -- with 20 timed steps that one-off showed as a 1.8-ms first step)
    timers.enabled = args.event_every > 0
    step()
    timers.enabled = False
    for k_ in timers.pairs:
        timers.pairs[k_] = []
    # A fixed DURATION of the workload ahead of the timed steps, whatever --warmup says: the driver runs
    # `--steps 20 --warmup 5` (25 ms of GPU time in all), and a box whose clocks have not settled reads several
    # per cent low.  It runs HERE, directly in front of the timed region -- behind the
    # statistics, the staged-entry count and the collector pass above, which idle the GPU for tens of milliseconds (in its
    # first place, ahead of them, the driver-form run still read 1.10 ms against 1.02: per-step times falling from 1.09 to
    # 1.00 over the 20 timed steps).  The number of extra steps is derived from the rate of ten more steps (the
    # first ones carry one-off costs) and agreed across ranks (every step carries collectives under data parallelism).
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    for _ in range(10):
        out = step()
    torch.cuda.synchronize()
    est_ms = 1e3 * (time.perf_counter() - t_w) / 10
    if dp:
        tt_ = torch.tensor([est_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        est_ms = float(tt_.item())
    t_w = time.perf_counter()
    if dp:  # a count every rank agrees on (with margin: the estimate runs slower than the settled rate)
        fixed_warmup_steps = int(min(4000, max(0, np.ceil(1.5e3 * FIXED_WARMUP_SECONDS / max(est_ms, 1e-3)))))
        for _ in range(fixed_warmup_steps):
            out = step()
        torch.cuda.synchronize()
    else:  # by the clock
        fixed_warmup_steps = 0
        while time.perf_counter() - t_w < FIXED_WARMUP_SECONDS and fixed_warmup_steps < 20000:
            for _ in range(20):
                out = step()
            torch.cuda.synchronize()
            fixed_warmup_steps += 20
    fixed_warmup_s = time.perf_counter() - t_w
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # (created before the barrier below)
    barrier()
    bracketed_steps = 0
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        # per-kernel HIP events on every `event_every`-th timed step: a pair of event
        # records per native call costs ~4 us of GPU idle time (18 pairs: 5 % of a step)
        timers.enabled = args.event_every > 0 and i % args.event_every == 0
        bracketed_steps += int(timers.enabled)
        step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    timers.enabled = False
    calib = calibration(dev) if rank == 0 else None
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])

    # The product path builds a view's tile lists on a side stream, next to the caller's own work, and composites in
    # the same native call where it can: HIP events around "one stage" do not isolate anything there.  The per-stage
    # table therefore comes from a few extra UNTIMED steps that run the stages one after the other on one stream
    # (GSR_SPECULATE=0, `one_call` = 0: the same kernels, the same inputs); the compositing backward -- the
    # dominant kernel of `roofline` -- is bracketed inside the timed region as well, and that is the figure used.
    pairs_timed, iso_steps = timers.pairs, 0
    if args.event_every > 0:  # (every rank: a step carries the gradient exchange's collectives)
        from rasterizer import rasterize as _Riso

        from rasterizer.cuda import _tuning as _Tune

        _Riso._speculation_mode()
        saved_mode, saved_over = _Riso._spec_knobs["mode"], _Tune.overrides()
        _Riso._spec_knobs["mode"] = "0"
        _Tune.set_overrides(dict(saved_over, one_call=0))
        timers.pairs = {k: [] for k in pairs_timed}
        try:
            step()
            timers.enabled = True
            for _ in range(8):
                step()
                iso_steps += 1
        finally:
            timers.enabled = False
            _Riso._spec_knobs["mode"] = saved_mode
            _Tune.set_overrides(saved_over)
        torch.cuda.synchronize()
        pairs_iso, timers.pairs = timers.pairs, pairs_timed
        for k_, v_ in pairs_iso.items():  # stages the timed region could not isolate: the sequential samples
            if k_ not in ("raster_bwd",) or not pairs_timed[k_]:
                pairs_timed[k_] = v_
        if pairs_iso["raster_bwd"] and pairs_timed["raster_bwd"] is not pairs_iso["raster_bwd"]:
            bracketed_bwd = bracketed_steps
        else:
            bracketed_bwd = iso_steps

    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_per_step = 1e3 * elapsed / args.steps
    pixels = W * H
    value = world * pixels / (elapsed / args.steps) / 1e6

    # The same K steps again with the host blocked where the UNCHANGED models block it (render_view `caller_syncs`):
    # `if (self.radii).sum() == 0` and `assert (num_tiles_hit > 0).any()` (vanilla_gs.py:784,811); and once more with
    # the intrinsics' `.item()` read-backs ahead of the projection as well (:736-740,772-773), which drain the stream
    # at the head of every view.  `value` above is what a caller without read-backs gets; these are what
    # `gs-train` gets with the models as they are.
    synced = {}
    for mode, key in ((True, "caller_syncs"), ("camera", "caller_and_camera_syncs")):
        if args.no_synced_regions:
            synced[key] = (None, None)
            continue
        for _ in range(min(3, args.warmup)):
            step(mode)
        gc.collect()
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            step(mode)
        barrier()
        el = time.perf_counter() - ts
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        synced[key] = (1e3 * el / args.steps, world * pixels / (el / args.steps) / 1e6)
    gpu_results = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dp:
        # one more (untimed) step whose results go to the host: what `parity_vs_oracle` checks
        for p in plist:
            p.grad = None
        o = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"], params["sh_coeffs"],
                        camt, bg, deg_use, clamp_rgb=False, retain_xys_grad=True)
        torch.autograd.backward([o["rgb"], o["alpha"]], [v_img, v_alpha[..., None]])
        torch.cuda.synchronize()
        c = lambda t_: t_.detach().cpu().numpy()  # noqa: E731
        gpu_results = {"rgb": c(o["rgb"]), "alpha": c(o["alpha"])[..., 0], "xys": c(o["xys"]), "radii": c(o["radii"]),
                       "grads": dict({k: c(params[k].grad) for k in ("means3d", "scales", "quats", "opacities", "sh_coeffs")},
                                     xys=c(o["xys"].grad))}
        del o
    allreduce_bytes = exchange.bytes_last if dp else None
    if dp:
        # the timed job is over: the other ranks leave (and free their GPUs) while rank 0 runs the
        # training leg in a fresh process group of its own and then prints the line
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        kern_ms = timers.summary_ms()
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        K = S.num_sh_bases(deg)
        # per-kernel bytes for what one launch processes (the lists as built: `list_entries`);
        # the end-to-end figure prices the job as SURVEY 8(d) defines it (the reference's lists)
        alg = S.algorithmic_bytes(N, list_entries, pixels, tiles, K)
        alg_job = S.algorithmic_bytes(N, num_intersects, pixels, tiles, K)
        # the compositing kernels are priced on the list entries they really stage (a tile stops
        # once all its pixels are saturated), not on the whole lists: 40 B (fwd) / 76 B (bwd) per
        # entry + the per-pixel images (40 I + 20 P, 76 I + 24 P)
        alg["raster_fwd"] = 40 * staged_fwd + 20 * pixels
        alg["raster_bwd"] = 76 * staged_bwd + 24 * pixels
        # the stages as BUILT move other bytes than the reference's scan / map / sort / bin-edges
        # organisation SURVEY prices: per-stage figures use these
        alg.update(S.built_pipeline_bytes(N, list_entries, tiles))
        if not timers.pairs["count_reach"] and timers.pairs["depth_order"]:
            alg["depth_order"] += alg["count_reach"]  # one call wrote the records too (gsr_reach_records_depth_order)
        step_ms_by_stage = timers.total_ms(max(iso_steps, 1))
        if iso_steps and bracketed_bwd != iso_steps:  # (the backward's samples come from the timed region)
            step_ms_by_stage["raster_bwd"] *= max(iso_steps, 1) / max(bracketed_bwd, 1)
        # the dominant stage is the one with the most time per STEP (calls x mean), not per call
        dominant = max(step_ms_by_stage, key=lambda k: step_ms_by_stage[k])
        ach = alg[dominant] / (kern_ms[dominant] * 1e-3) / 1e9 if kern_ms[dominant] > 0 else 0.0
        roofline = {
            "kernel": dominant, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
            "algorithmic_bytes": alg[dominant], "kernel_ms": round(kern_ms[dominant], 4),
            "launches_per_step": round(step_ms_by_stage[dominant] / kern_ms[dominant], 2) if kern_ms[dominant] > 0 else 0,
            "ms_per_step_in_kernel": round(step_ms_by_stage[dominant], 4),
            "staged_list_entries": {"raster_fwd": staged_fwd, "raster_bwd": staged_bwd, "lists": list_entries},
        }
        # HBM traffic and VALU occupancy of the dominant kernel, measured now: separate rocprofv3
        # counter passes over a short run of this same command (skipped with --no-pmc / N > 1)
        stage_traffic = None
        if world == 1 and not args.no_pmc:
            sub = [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)]
            for flag in ("--steps", "--warmup", "--event-every", "--gpus", "--train-iters"):
                while flag in sub:
                    i = sub.index(flag)
                    del sub[i:i + 2]
            sub += ["--steps", "3", "--warmup", "2", "--event-every", "0", "--no-cpu-baseline", "--no-pmc",
                    "--train-iters", "0"] + ([] if "--no-synced-regions" in sub else ["--no-synced-regions"])
            sub_steps = 2 + 1 + 3  # warm-up, the staged-entry count step, timed (fallback only: see pmc_steps)
            f = pmc_pass(["FETCH_SIZE"], sub)
            w = pmc_pass(["WRITE_SIZE"], sub)
            # KB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
            to_bytes = lambda fk, wk: int((2 * fk + wk) * 1024)
            steps_f, steps_w = pmc_steps(f, "FETCH_SIZE", sub_steps), pmc_steps(w, "WRITE_SIZE", sub_steps)
            fl, _ = pmc_stage(f, dominant, "FETCH_SIZE", steps_f)
            wl, _ = pmc_stage(w, dominant, "WRITE_SIZE", steps_w)
            if fl is not None and wl is not None:
                roofline["traffic"] = to_bytes(fl, wl)  # per launch of the dominant kernel
                roofline["traffic_raw"] = {"FETCH_SIZE_KB": round(fl, 1), "WRITE_SIZE_KB": round(wl, 1)}
            stage_traffic = {}
            for st in STAGE_KERNELS:
                _, fs = pmc_stage(f, st, "FETCH_SIZE", steps_f)
                _, ws = pmc_stage(w, st, "WRITE_SIZE", steps_w)
                if fs is not None and ws is not None:
                    stage_traffic[st] = to_bytes(fs, ws)  # all launches of the stage, per step
            q = pmc_pass(["SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"], sub)
            # both counters over ALL launches of the stage (a two-pass step launches the kernel twice with
            # different work): instructions and resident cycles of the same set of launches
            _, qi = pmc_stage(q, dominant, "SQ_INSTS_VALU", 1)
            _, qa = pmc_stage(q, dominant, "GRBM_GUI_ACTIVE", 1)
            if qi is not None and qa:
                # the compositing kernels are VALU-issue bound, not HBM bound:
                # wave64 VALU instructions x 4 cycles / SIMD-cycles the kernel was resident
                simd_cycles = qa / 8.0 * 1024.0
                roofline["valu_busy"] = round(qi * 4.0 / simd_cycles, 3)
                roofline["limiter"] = "VALU issue (SQ_INSTS_VALU x 4 cycles / SIMD-cycles resident, this run)"
                cpi = VALU_CYCLES_PER_INSTRUCTION.get(dominant)
                if cpi:
                    # ... with every instruction priced at its MEASURED issue cost instead of a flat 4 cycles
                    roofline["valu_pipe_busy"] = round(qi * cpi / simd_cycles, 3)
                    roofline["valu_pipe_busy_what"] = (f"SQ_INSTS_VALU x {cpi} cycles (the loop's mean issue cost per "
                                                       "instruction by the measured per-class table) / SIMD-cycles resident")
            qi_launch, _ = pmc_stage(q, dominant, "SQ_INSTS_VALU", 1)
            if qi_launch is not None and kern_ms[dominant] > 0:
                # the roofline the compositing kernels ARE at: wave64 VALU instructions of one launch x 64 lanes over
                # the launch's duration (HIP events, this run) against the non-packed fp32 issue peak
                lane_ops = qi_launch * 64.0 / (kern_ms[dominant] * 1e-3)
                roofline["valu"] = {"lane_ops_per_s": round(lane_ops / 1e12, 2), "unit": "T lane-ops/s",
                                    "peak": round(VALU_PEAK_LANE_OPS / 1e12, 1),
                                    "peak_what": "256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz: the plain class (add / mul / fmac, "
                                                 "~2.4 cycles per wave64 instruction measured); v_fma 2.7, cmp / cndmask / min / "
                                                 "DPP 4.2, exp / rcp / permlane_swap 8.2 -- see valu_pipe_busy",
                                    "frac": round(lane_ops / VALU_PEAK_LANE_OPS, 3),
                                    "valu_instructions_per_launch": int(qi_launch)}
        # every kernel's own fraction, and the end-to-end figure from SURVEY 8(d)
        # per stage: mean ms of one call, calls per step, its algorithmic bytes per call (as built),
        # the rate that gives, and -- with the counter passes -- the HBM bytes all its launches moved per step
        per_kernel = {}
        for k in kern_ms:
            calls = step_ms_by_stage[k] / kern_ms[k] if kern_ms[k] > 0 else 0.0
            gbps = alg[k] / (kern_ms[k] * 1e-3) / 1e9 if kern_ms[k] > 0 else 0.0
            per_kernel[k] = {"ms": round(kern_ms[k], 4), "calls_per_step": round(calls, 2),
                             "algorithmic_bytes": int(alg[k]), "GBps": round(gbps, 1),
                             "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBS, 4)}
            if stage_traffic and k in stage_traffic:
                per_kernel[k]["traffic_per_step"] = stage_traffic[k]
        end_to_end = alg_job["total"] / (ms_per_step * 1e-3) / 1e9
        # the same rate on the bytes the pipeline AS BUILT must move (every stage's own figure x its calls per step:
        # exact-reach lists, staged entries only, two-round lists where they apply): this one cannot exceed the peak
        built_bytes = sum(alg[k] * (step_ms_by_stage[k] / kern_ms[k] if kern_ms[k] > 0 else 0.0) for k in kern_ms)
        end_to_end_built = built_bytes / (ms_per_step * 1e-3) / 1e9

        cpu = cpu1 = parity = None
        if world == 1 and not args.no_cpu_baseline:
            kept = {} if (deg_use == deg and not args.render_depth) else None
            cpu, _ = cpu_baseline(sc, cam, bg_np, v_img_np, v_alpha_np, deg, keep=kept)
            cpu["value"] = round(cpu["value"], 4)
            if kept:
                try:
                    parity = parity_vs_oracle(gpu_results, kept)
                except Exception as e:  # the line must survive
                    parity = {"error": repr(e)}
                kept.clear()
            if not args.no_cpu_one_thread:
                cpu1 = cpu_baseline_one_thread(sc, cam, bg_np, v_img_np, v_alpha_np, deg)
                cpu1["value"] = round(cpu1["value"], 4)
        res_name = "1080p" if (W, H) == (1920, 1080) else ("4K" if (W, H) == (3840, 2160) else f"{W}x{H}")
        n_name = f"{N // 1_000_000}M" if N % 1_000_000 == 0 else (f"{N // 1000}k" if N % 1000 == 0 else str(N))

        line = {
            "metric": f"raster fwd+bwd Mpix/s @{res_name} ({n_name} Gaussians, SH{deg})"
                      + ("; train iters/s in `train`" if args.train_iters > 0 else ""),
            "value": round(value, 2),
            "value_normalised": (round(value * CALIBRATION_REFERENCE_VALU_TOPS / calib["valu_Tops"], 2)
                                 if calib and calib.get("valu_Tops") else None),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            # the same steps with the unchanged models' host read-backs in the caller (see `synced` above)
            "value_with_caller_syncs": _rnd(synced["caller_syncs"][1], 2),
            "ms_per_step_with_caller_syncs": _rnd(synced["caller_syncs"][0], 4),
            "caller_syncs_gap": (None if synced["caller_syncs"][0] is None
                                 else round(synced["caller_syncs"][0] / ms_per_step - 1.0, 4)),
            "value_with_caller_and_camera_syncs": _rnd(synced["caller_and_camera_syncs"][1], 2),
            "ms_per_step_with_caller_and_camera_syncs": _rnd(synced["caller_and_camera_syncs"][0], 4),
            "caller_syncs_what": ("caller_syncs: `if (self.radii).sum() == 0` + `assert (num_tiles_hit > 0).any()` "
                                  "(vanilla_gs.py:784,811) block the host twice per view; caller_and_camera_syncs: plus "
                                  "the eight intrinsics read-backs ahead of the projection (:736-740,772-773), the first "
                                  "of which drains the stream; `value` = no read-backs"),
            "ms_per_step_median": round(float(np.median(step_ms)), 4),
            "ms_per_step_p10_p90": [round(float(np.percentile(step_ms, 10)), 4),
                                    round(float(np.percentile(step_ms, 90)), 4)],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{N} Gaussians of a TRAINED model ({os.path.basename(args.scene[4:])}; view {args.ply_view} of "
                             f"the training orbit), " if args.scene.startswith("ply:") else
                             (f"{N} Gaussians of the held-out family '{args.scene}' (harness.scene.make_heldout_scene), "
                              if args.scene in S.HELDOUT_KINDS else
                              f"{N} random Gaussians (SURVEY 8d, seed 42, scales log-U[{args.scale_lo},{args.scale_hi}]), ")) +
                            f"SH degree {deg}, {W}x{H}, block 16, fwd+bwd through the rasterizer autograd API"
                            + ((" + depth image from the same compositing pass (gs_fused)" if args.fused_depth
                               else " + differentiable depth pass") if args.render_depth else "")
                            + (", deterministic backward" if args.deterministic else ""),
                "intersections_per_gaussian": round(num_intersects / N, 2),
                "gaussians": N, "visible": n_visible, "intersections": num_intersects,
                "list_entries": list_entries,
                "mean_gaussians_per_tile": round(num_intersects / tiles, 1),
                "tile_list_length": tile_hist, "scene": args.scene, "two_round_lists": two_round,
                "cpu_bind": cpu_bind,
                # (inside `config` so that the driver's `parsed` keeps them)
                "timing": {"ms_per_step_median": round(float(np.median(step_ms)), 4),
                           "ms_per_step_p10_p90": [round(float(np.percentile(step_ms, 10)), 4),
                                                   round(float(np.percentile(step_ms, 90)), 4)],
                           "fixed_warmup_steps": fixed_warmup_steps, "fixed_warmup_s": round(fixed_warmup_s, 3),
                           "ms_per_step_each": ([round(float(v), 4) for v in step_ms] if args.steps <= 40 else None),
                           "what": f"--warmup steps, then {FIXED_WARMUP_SECONDS} s of the same workload untimed, then "
                                   "exactly --steps timed steps (per-step figures: HIP events between steps)"},
                "calibration": calib,
                "value_normalised": (round(value * CALIBRATION_REFERENCE_VALU_TOPS / calib["valu_Tops"], 2)
                                     if calib and calib.get("valu_Tops") else None),
                "parallelism": (f"dp{world} (per-view; "
                                + ("geometry gradients all-reduced (one flat message), SH gradient formed on every rank "
                                   "from the all-gathered 12-byte colour cotangents (gsr_sh_backward_views), both started "
                                   "from autograd hooks; " if sh_views else
                                   "per-parameter all-reduce started from autograd hooks, overlapping "
                                   "the backward; active SH bands only; ")
                                + ("averaged in the collective, RCCL)" if args.backend == "nccl" else
                                   f"{args.backend}: ranks share GPUs, not a measurement configuration)"))
                               if world > 1 else "single",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity_vs_oracle": parity,
            "cpu_baseline_one_thread_60k_gaussian_subset": cpu1,
            "kernels": per_kernel,
            "kernel_events": (f"raster_bwd: HIP events around the native call on every {max(args.event_every, 1)}th TIMED step; the "
                              "other stages: HIP events over 8 extra untimed steps that run the stages one after the other "
                              "on one stream (GSR_SPECULATE=0, one_call = 0) -- in the timed steps the tile lists are built "
                              "on a side stream next to the SH evaluation, which no pair of events isolates"),
            "end_to_end_algorithmic_GBps": round(end_to_end, 1),
            "end_to_end_built_GBps": round(end_to_end_built, 1),
            "end_to_end_note": ("end_to_end_algorithmic prices SURVEY 8d's formula (748 N + 160 I + 44 P with the reference's "
                                "3-sigma intersections I); end_to_end_built prices what the stages as built must move"
                                + ("; the former exceeds the 8000 GB/s peak here because the lists this pipeline builds are "
                                   "shorter than the reference's (work legitimately not done) -- it is not a roofline figure, "
                                   "the built one is" if end_to_end > HBM_PEAK_GBS else "")),
            # N > 1: the exchange of the 59-float/Gaussian gradient as rank 0 sees it
            "allreduce_ms": (round(float(np.mean([a.elapsed_time(b) for a, b in comm_events])), 4)
                             if comm_events else None),
            "allreduce_ms_note": "exposed part: from the end of the queued backward to the last collective" if comm_events else None,
            "allreduce_bytes": allreduce_bytes,
        }
        # the second half of BASELINE's metric: config 3 (N = 1) / config 4 (N > 1) training, timed inside
        # this run (a fresh process: its failure cannot cost the line above)
        line["train"] = train_record(args, world) if args.train_iters > 0 else None
        tr = line["train"]
        g_ = lambda d_, *ks: (g_(d_.get(ks[0]), *ks[1:]) if len(ks) > 1 else d_.get(ks[0])) if isinstance(d_, dict) else None  # noqa: E731
        # SCALARS at the top level of `config` / `roofline`: the driver's record keeps those and drops nested objects
        #  -- what tells box from code, and both halves of BASELINE's metric, survive in `parsed`
        cfg_ = line["config"]
        cfg_["ms_per_step_median"] = round(float(np.median(step_ms)), 4)
        cfg_["ms_per_step_p10"] = round(float(np.percentile(step_ms, 10)), 4)
        cfg_["ms_per_step_p90"] = round(float(np.percentile(step_ms, 90)), 4)
        cfg_["calib_valu_Tops"] = g_(calib, "valu_Tops")
        cfg_["calib_copy_GBps"] = g_(calib, "copy_GBps")
        cfg_["parity_image_max_abs"] = g_(parity, "img_max_abs_stable")
        cfg_["parity_grad_max_rel"] = (max(parity["grad_max_abs_over_max_ref"].values())
                                       if isinstance(parity, dict) and parity.get("grad_max_abs_over_max_ref") else None)
        cfg_["parity_meets"] = (bool(all(parity["meets"].values())) if isinstance(parity, dict) and "meets" in parity else None)
        if isinstance(roofline.get("valu"), dict):
            roofline["valu_frac"] = roofline["valu"].get("frac")
            roofline["valu_lane_Tops"] = roofline["valu"].get("lane_ops_per_s")
        if isinstance(tr, dict) and "error" not in tr:
            r480 = g_(tr, "with_caller_syncs", "phase_ms_median_by_resolution")
            keys = sorted(r480, key=lambda k_: int(k_.split("x")[0])) if isinstance(r480, dict) else []
            for k_ in keys:  # (e.g. render_480x270_ms: the render phase with the models' read-backs, per stage of the schedule)
                cfg_[f"render_{k_}_ms"] = g_(r480, k_, "render")
            r480n = g_(tr, "phase_ms_median_by_resolution")
            for k_ in (sorted(r480n, key=lambda k2: int(k2.split("x")[0])) if isinstance(r480n, dict) else []):
                cfg_[f"render_{k_}_ms_no_readbacks"] = g_(r480n, k_, "render")
            if keys:  # (the name VERDICT r5 asks for: the schedule's lowest resolution, 480 x 270 on config 3)
                cfg_["render_480_ms"] = g_(r480, keys[0], "render")
            cfg_["render_480_slow_share"] = g_(tr, "with_caller_syncs", "render_slow_share_lowest_resolution")
            tp = g_(tr, "trained_raster", "parity_vs_oracle")
            cfg_.update({
                "config3_iters_per_s": tr.get("iters_per_s"),
                "config3_with_caller_syncs_iters_per_s": tr.get("iters_per_s_with_caller_syncs"),
                "config3_unchanged_caller_iters_per_s": tr.get("iters_per_s_unchanged_caller"),
                "config3_full_resolution_iters_per_s": g_(tr, "full_resolution_from_step_0", "iters_per_s"),
                "config3_one_op_iters_per_s": g_(tr, "one_op_path", "iters_per_s"),
                "fixed_1m_iters_per_s": g_(tr, "fixed_1m", "iters_per_s"),
                "refined_1m_iters_per_s": g_(tr, "refined_1m", "iters_per_s"),
                "cogs_3m_4k_iters_per_s": g_(tr, "cogs_3m_4k", "iters_per_s"),
                "cogs_3m_4k_peak_memory_GB": g_(tr, "cogs_3m_4k", "peak_memory_GB"),
                "trained_raster_ms": g_(tr, "trained_raster", "ms"),
                "trained_raster_fwd_ms": g_(tr, "trained_raster", "raster_fwd_ms"),
                "trained_raster_bwd_ms": g_(tr, "trained_raster", "raster_bwd_ms"),
                "trained_valu_busy": g_(tr, "trained_raster", "valu_busy"),
                "trained_traffic_bytes": g_(tr, "trained_raster", "traffic"),
                "trained_parity_image_max_abs": g_(tp, "img_max_abs_stable"),
                "trained_parity_grad_max_rel": (max(tp["grad_max_abs_over_max_ref"].values())
                                                if isinstance(tp, dict) and tp.get("grad_max_abs_over_max_ref") else None),
                "trained_parity_meets": (bool(all(tp["meets"].values())) if isinstance(tp, dict) and "meets" in tp else None),
            })
            # (the nested digest of rounds 4-5, kept for continuity; the driver drops it)
            line["config"]["train_digest"] = {
                "config3_iters_per_s": tr.get("iters_per_s"),
                "config3_with_caller_syncs_iters_per_s": tr.get("iters_per_s_with_caller_syncs"),
                "fixed_1m_iters_per_s": g_(tr, "fixed_1m", "iters_per_s"),
                "refined_1m_iters_per_s": g_(tr, "refined_1m", "iters_per_s"),
                "cogs_3m_4k": {k_: g_(tr, "cogs_3m_4k", k_) for k_ in ("iters_per_s", "gaussians_end", "peak_memory_GB",
                                                                         "list_overflow_views", "psnr", "error")},
                "trained_raster": {k_: g_(tr, "trained_raster", k_) for k_ in ("ms", "raster_fwd_ms", "raster_bwd_ms",
                                                                                 "valu_busy", "traffic", "error")},
            }
        elif isinstance(tr, dict):
            cfg_["train_error"] = str(tr.get("error"))[:200]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
