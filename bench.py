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
    ("bin_sorted", "bin_sorted"),
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
        return {k: (float(np.mean([a.elapsed_time(b) for a, b in v])) if v else 0.0)
                for k, v in self.pairs.items()}


def cpu_baseline(sc, cam, bg, v_img, v_alpha, deg):
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
                         sc["opacities"], bg)
    t1 = time.perf_counter()
    vxy, vconic, vcol, vop = O.rasterize_backward(
        cam.height, cam.width, 16, r["gaussian_ids_sorted"], r["tile_bins"], r["xys"], r["conics"], rgbs,
        sc["opacities"], bg, r["final_Ts"], r["final_idx"], v_img, v_alpha)
    O.compute_sh_backward(n, deg, deg, dirs, (vcol * (sh + 0.5 > 0)).astype(np.float32))
    zeros = np.zeros(n, np.float32)
    O.project_gaussians_backward(n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3],
                                 cam.projmat, cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width,
                                 r["cov3d"], r["radii"], r["conics"], r["compensation"], vxy, zeros, vconic,
                                 zeros)
    t2 = time.perf_counter()
    pix = cam.width * cam.height
    return {
        "value": pix / (t2 - t0) / 1e6,
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": f"1 full fwd+bwd of the same workload ({n} Gaussians, {cam.width}x{cam.height}); "
                  f"fwd {t1 - t0:.2f}s bwd {t2 - t1:.2f}s; host has {os.cpu_count()} logical CPUs",
    }, r["num_intersects"]


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
    res["sample"] = (f"first {n} Gaussians of the workload ({I} reference list entries) at {cam.width}x{cam.height}, one "
                     f"thread; " + res["sample"].split("; ", 1)[1])
    return res


def pmc_pass(counters, argv, kernel_substr, timeout_s=240):
    """One separate `rocprofv3 --kernel-trace --pmc <counters>` pass over a short run of this
    script (MI355X_MICROARCH.md, HBM section: counters in their own pass, kernel trace only).
    -> {counter: mean per launch of the kernels whose name contains `kernel_substr`} or None."""
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
                if kernel_substr in row["Kernel_Name"]:
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        return {k: float(np.mean(v)) for k, v in vals.items()} or None
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
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
    ap.add_argument("--deterministic", action="store_true",
                    help="compositing backward with a fixed summation order (gsr_rasterize_backward_det) instead of "
                         "float atomics")
    ap.add_argument("--scene", default="uniform", choices=["uniform", "longtail"],
                    help="longtail: 10 %% of the tiles hold ~10x the list depth (clustered Gaussians)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
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
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = args.backend
    if backend == "auto":
        backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    args.backend = backend
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    cam0 = S.make_camera(W, H)
    sc = S.make_scene(N, cam0, sh_degree=deg, seed=42, scale_lo=args.scale_lo, scale_hi=args.scale_hi,
                      longtail=args.scene == "longtail")
    # rank r looks at the same cloud from a slightly different direction
    cam = cam0 if rank == 0 else S.make_camera(W, H, yaw=0.02 * rank, pitch=0.01 * (rank % 3))
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
                                average=True).attach()
    exchange.active_rows["sh_coeffs"] = (deg_use + 1) ** 2

    def step():
        for p in plist:
            p.grad = None
        out = render_view(params["means3d"], params["scales"], params["quats"], params["opacities"],
                          params["sh_coeffs"], camt, bg, deg_use, clamp_rgb=False, render_depth=args.render_depth,
                          fused_depth=args.fused_depth)
        if args.render_depth:
            torch.autograd.backward([out["rgb"], out["alpha"], out["depth"]],
                                    [v_img, v_alpha[..., None], v_alpha[..., None]])
        else:
            torch.autograd.backward([out["rgb"], out["alpha"]], [v_img, v_alpha[..., None]])
        if world > 1:
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
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    num_intersects = int(out["num_tiles_hit"].sum().item())  # the reference's lists (3-sigma boxes)
    from rasterizer import rasterize as _R
    list_entries = int(_R._bin_cache["value"][0])  # what the kernels walk (dead pairs left out)
    n_visible = int((out["radii"] > 0).sum().item())
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

    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        # per-kernel HIP events on every `event_every`-th timed step: a pair of event
        # records per native call costs ~4 us of GPU idle time (18 pairs: 5 % of a step)
        timers.enabled = args.event_every > 0 and i % args.event_every == 0
        step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    timers.enabled = False
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])

    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_per_step = 1e3 * elapsed / args.steps
    pixels = W * H
    value = world * pixels / (elapsed / args.steps) / 1e6

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
        dominant = max(kern_ms, key=lambda k: kern_ms[k])
        ach = alg[dominant] / (kern_ms[dominant] * 1e-3) / 1e9 if kern_ms[dominant] > 0 else 0.0
        roofline = {
            "kernel": dominant, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
            "algorithmic_bytes": alg[dominant], "kernel_ms": round(kern_ms[dominant], 4),
            "staged_list_entries": {"raster_fwd": staged_fwd, "raster_bwd": staged_bwd, "lists": list_entries},
        }
        # HBM traffic and VALU occupancy of the dominant kernel, measured now: separate rocprofv3
        # counter passes over a short run of this same command (skipped with --no-pmc / N > 1)
        kn = {"raster_bwd": "raster_bwd_tile16_kernel", "raster_fwd": "raster_fwd_tile16_kernel",
              "sh_fwd": "sh16_fwd_kernel", "sh_bwd": "sh16_bwd_kernel", "project_fwd": "project_fwd_kernel",
              "project_bwd": "project_bwd_kernel"}.get(dominant)
        if world == 1 and not args.no_pmc and kn is not None:
            sub = [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)]
            for flag in ("--steps", "--warmup", "--event-every", "--gpus"):
                while flag in sub:
                    i = sub.index(flag)
                    del sub[i:i + 2]
            sub += ["--steps", "3", "--warmup", "2", "--event-every", "0", "--no-cpu-baseline", "--no-pmc"]
            f = pmc_pass(["FETCH_SIZE"], sub, kn)
            w = pmc_pass(["WRITE_SIZE"], sub, kn)
            if f and w and "FETCH_SIZE" in f and "WRITE_SIZE" in w:
                # KB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
                roofline["traffic"] = int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024)
                roofline["traffic_raw"] = {"FETCH_SIZE_KB": round(f["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(w["WRITE_SIZE"], 1)}
            q = pmc_pass(["SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"], sub, kn)
            if q and "SQ_INSTS_VALU" in q and q.get("GRBM_GUI_ACTIVE", 0) > 0:
                # the compositing kernels are VALU-issue bound, not HBM bound:
                # wave64 VALU instructions x 4 cycles / SIMD-cycles the kernel was resident
                simd_cycles = q["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
                roofline["valu_busy"] = round(q["SQ_INSTS_VALU"] * 4.0 / simd_cycles, 3)
                roofline["limiter"] = "VALU issue (SQ_INSTS_VALU x 4 cycles / SIMD-cycles resident, this run)"
        # every kernel's own fraction, and the end-to-end figure from SURVEY 8(d)
        per_kernel = {
            k: {"ms": round(kern_ms[k], 4), "GBps": round(alg[k] / (kern_ms[k] * 1e-3) / 1e9, 1) if kern_ms[k] > 0 else 0.0}
            for k in kern_ms
        }
        end_to_end = alg_job["total"] / (ms_per_step * 1e-3) / 1e9

        cpu = cpu1 = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, _ = cpu_baseline(sc, cam, bg_np, v_img_np, v_alpha_np, deg)
            cpu["value"] = round(cpu["value"], 4)
            cpu1 = cpu_baseline_one_thread(sc, cam, bg_np, v_img_np, v_alpha_np, deg)
            cpu1["value"] = round(cpu1["value"], 4)
        res_name = "1080p" if (W, H) == (1920, 1080) else ("4K" if (W, H) == (3840, 2160) else f"{W}x{H}")
        n_name = f"{N // 1_000_000}M" if N % 1_000_000 == 0 else (f"{N // 1000}k" if N % 1000 == 0 else str(N))

        line = {
            "metric": f"raster fwd+bwd Mpix/s @{res_name} ({n_name} Gaussians, SH{deg})",
            "value": round(value, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(float(np.median(step_ms)), 4),
            "ms_per_step_p10_p90": [round(float(np.percentile(step_ms, 10)), 4),
                                    round(float(np.percentile(step_ms, 90)), 4)],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{N} random Gaussians (SURVEY 8d, seed 42, scales log-U[{args.scale_lo},{args.scale_hi}]), "
                            f"SH degree {deg}, {W}x{H}, block 16, fwd+bwd through the rasterizer autograd API"
                            + ((" + depth image from the same compositing pass (gs_fused)" if args.fused_depth
                               else " + differentiable depth pass") if args.render_depth else "")
                            + (", deterministic backward" if args.deterministic else ""),
                "intersections_per_gaussian": round(num_intersects / N, 2),
                "gaussians": N, "visible": n_visible, "intersections": num_intersects,
                "list_entries": list_entries,
                "mean_gaussians_per_tile": round(num_intersects / tiles, 1),
                "tile_list_length": tile_hist, "scene": args.scene,
                "parallelism": (f"dp{world} (per-view; per-parameter all-reduce started from autograd hooks, overlapping "
                                f"the backward; active SH bands only; "
                                + ("averaged in the collective, RCCL)" if args.backend == "nccl" else
                                   f"{args.backend}: ranks share GPUs, not a measurement configuration)"))
                               if world > 1 else "single",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_one_thread": cpu1,
            "kernels": per_kernel,
            "kernel_events": f"HIP events around each native call on every {args.event_every}th timed step" if args.event_every > 1 else "HIP events around each native call on every timed step",
            "end_to_end_algorithmic_GBps": round(end_to_end, 1),
            # N > 1: the exchange of the 59-float/Gaussian gradient as rank 0 sees it
            "allreduce_ms": (round(float(np.mean([a.elapsed_time(b) for a, b in comm_events])), 4)
                             if comm_events else None),
            "allreduce_ms_note": "exposed part: from the end of the queued backward to the last collective" if comm_events else None,
            "allreduce_bytes": exchange.bytes_last if world > 1 else None,
        }
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
