#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 5 from gpurun_out/bench_r04_*.json (or profiles/)."""
import glob
import json
import os
import sys

d0 = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for f in sorted(glob.glob(os.path.join(d0, "bench_r04_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e)
        continue
    name = os.path.basename(f)[len("bench_r04_"):-5]
    c = d["config"]
    print(f"| {name} | {c['gaussians']} | I {c['intersections']} -> {c['list_entries']} | {d['ms_per_step']} (median {d['ms_per_step_median']}) | "
          f"{d['value']} | synced {d.get('ms_per_step_with_caller_syncs')} / camera {d.get('ms_per_step_with_caller_and_camera_syncs')} | "
          f"two-round {c.get('two_round_lists')} |")
    k = d["kernels"]
    print("    kernels:", {n: (v["ms"], v["calls_per_step"], v["frac_of_hbm_peak"]) for n, v in k.items() if v["ms"]})
    if name == "default":
        print("    roofline:", json.dumps(d["roofline"]))
        print("    parity:", json.dumps(d.get("parity_vs_oracle"))[:1200])
        print("    cpu:", d.get("cpu_baseline"), d.get("cpu_baseline_one_thread_60k_gaussian_subset"))
        t = d.get("train") or {}
        for key in ("iters_per_s", "iters_per_s_with_caller_syncs", "iters_per_s_unchanged_caller", "gaussians", "psnr",
                    "phase_ms_median_by_resolution", "list_overflow_views"):
            print("    train", key, json.dumps(t.get(key))[:600])
        for key in ("with_caller_syncs", "full_resolution_from_step_0", "unchanged_caller", "refined_1m", "one_op_path", "fixed_1m"):
            v = t.get(key) or {}
            print("    train", key, {kk: v.get(kk) for kk in ("iters_per_s", "gaussians_end", "gaussians_max", "psnr", "phase_ms_median")})
