#!/usr/bin/env python3
"""One fresh process of config 3's 480 x 270 phase: the render phase's per-iteration GPU time (HIP events around
every iteration), classified fast / slow, plus host-side time per iteration.  VERDICT r5 item 1: the cause of the
two modes (0.37 / 0.53 ms) before any kernel work.
    python tools/r06/mode480.py [--iters 600] [--syncs 0|1] [--tag name]
Variants ride in the environment (GSR_SPECULATE=0, GSR_POLL_YIELD=0, ...) or in --graph / --calib.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402  (puts the package on sys.path)
from harness.train import train  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=600)
ap.add_argument("--syncs", type=int, default=1)
ap.add_argument("--graph", type=int, default=0, help="the whole iteration as one HIP graph (ViewGraph)")
ap.add_argument("--fused", type=int, default=0, help="the one-op path (render_gaussians)")
ap.add_argument("--calib", type=int, default=0, help="run bench.calibration() before the leg (clock state)")
ap.add_argument("--tag", default="default")
ap.add_argument("--every", type=int, default=1)
args = ap.parse_args()

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if args.calib:
    bench.calibration(dev)
cfg = bench.config3(args.iters)
cfg.phase_every, cfg.phase_series = args.every, True
cfg.caller_syncs = bool(args.syncs)
cfg.eval_views = 1
if args.graph:
    cfg.use_graph = True
if args.fused:
    cfg.fused_render = True
import ctypes  # noqa: E402

_libc = ctypes.CDLL(None)


def _cpu():
    c = _libc.sched_getcpu()
    node = None
    try:
        node = [d for d in os.listdir(f"/sys/devices/system/cpu/cpu{c}") if d.startswith("node")][0]
    except Exception:
        pass
    return [c, node]


def round_trip_us(reps=400):
    """One blocking read-back as the unchanged models issue it: a small kernel and `.item()` (hipMemcpy D2H + wait);
    the median wall time of the pair with the queue otherwise empty -- the box / runtime, nothing of the rasterizer."""
    x = torch.ones(1024, device=dev)
    for _ in range(50):
        (x.sum() == 0).item()
    ts = []
    for _ in range(reps):
        a = time.perf_counter()
        (x.sum() == 0).item()
        ts.append(time.perf_counter() - a)
    ts = np.array(ts) * 1e6
    return {"p10": round(float(np.percentile(ts, 10)), 1), "p50": round(float(np.median(ts)), 1), "p90": round(float(np.percentile(ts, 90)), 1)}


def python_speed_us():
    """A fixed piece of pure-Python work (no GPU): how fast THIS process's interpreter thread runs right now."""
    best = 1e9
    for _ in range(5):
        a = time.perf_counter()
        x = 0
        for i in range(200_000):
            x += i & 7
        best = min(best, time.perf_counter() - a)
    return round(best * 1e6, 1)


rt_before = round_trip_us()
py_before = python_speed_us()
cpu_start = _cpu()
t0 = time.perf_counter()
res = train(cfg, dev, 0, 1)
wall = time.perf_counter() - t0
out = {"tag": args.tag, "syncs": args.syncs, "iters_per_s": round(res["iters_per_s"], 1), "wall_s": round(wall, 1),
       "gaussians_end": res["num_gaussians_end"]}
from rasterizer import rasterize as _RZ  # noqa: E402

out["counters"] = {k: v for k, v in _RZ.counters.items() if v}
out["cpu_start_end"] = [cpu_start, _cpu()]
out["round_trip_us_before_after"] = [rt_before, round_trip_us()]
out["python_200k_loop_us_before_after"] = [py_before, python_speed_us()]
out["affinity"] = len(os.sched_getaffinity(0))
try:
    out["gpu_numa_node"] = open("/sys/class/drm/card0/device/numa_node").read().strip()
except Exception:
    out["gpu_numa_node"] = None
ser = res.get("phase_ms_series")
if ser:
    a = np.array(ser)[:, 1:]  # render, loss, backward, optimizer
    a = a[len(a) // 10:]      # (the first tenth: warm-up, allocator growth)
    for i, k in enumerate(("render", "loss", "backward", "opt")):
        col = a[:, i]
        out[k] = {"p10": round(float(np.percentile(col, 10)), 4), "p50": round(float(np.median(col)), 4),
                  "p90": round(float(np.percentile(col, 90)), 4), "mean": round(float(col.mean()), 4)}
    r = a[:, 0]
    lo = np.percentile(r, 5)
    out["render_slow_share"] = round(float((r > 1.25 * lo).mean()), 3)   # samples > 1.25 x the leg's 5th percentile
    out["render_hist_ms"] = {f"{e:.2f}": int(c) for c, e in zip(*np.histogram(r, bins=np.arange(0.2, 1.01, 0.05)))}
    # medians of consecutive blocks of 50 samples: is the mode a property of the process or does it flip inside it?
    out["render_block_medians"] = [round(float(np.median(r[i:i + 50])), 3) for i in range(0, len(r) - 49, 50)]
print(json.dumps(out), flush=True)
