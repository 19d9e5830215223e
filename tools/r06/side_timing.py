#!/usr/bin/env python3
"""Per-iteration timing of the render phase's two streams from a rocprofv3 --kernel-trace CSV directory (config 3's
480 x 270 phase with the models' read-backs): when the side stream's list construction starts and ends relative to the
projection, how long the main stream's chain of read-back kernels takes, when the compositing starts.
    python tools/r06/side_timing.py <dir> [label]   -> one line of medians (us)"""
import csv
import glob
import statistics as st
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
K = list(csv.DictReader(open(f[0])))
for r in K:
    r["a"], r["b"], r["q"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
K.sort(key=lambda r: r["a"])
proj = [i for i, r in enumerate(K) if "project_fwd" in r["Kernel_Name"]]
if len(proj) < 40:
    print(label, "too few iterations", len(proj))
    sys.exit(0)
mainq = K[proj[0]]["q"]
rows = []
for n in range(len(proj) // 3, len(proj) - 1):
    i0, i1 = proj[n], proj[n + 1]
    it = K[i0:i1]
    pe = it[0]["b"]
    side = [r for r in it if r["q"] != mainq]
    comp = [r for r in it if r["q"] == mainq and ("raster_fwd" in r["Kernel_Name"])]
    if not comp:
        continue
    c0 = comp[0]
    before = [r for r in it if r["q"] == mainq and r["a"] < c0["a"] and r is not it[0]]
    main_busy = sum(r["b"] - r["a"] for r in before)
    rows.append({
        "side_n": len(side),
        "side_start": (side[0]["a"] - pe) / 1e3 if side else None,
        "side_end": (max(r["b"] for r in side) - pe) / 1e3 if side else None,
        "side_busy": sum(r["b"] - r["a"] for r in side) / 1e3 if side else None,
        "main_chain_end": (max(r["b"] for r in before) - pe) / 1e3 if before else 0.0,
        "main_chain_busy": main_busy / 1e3,
        "comp_start": (c0["a"] - pe) / 1e3,
        "comp_end": (comp[-1]["b"] - pe) / 1e3,
        "iter": (K[i1]["a"] - it[0]["a"]) / 1e3,
    })
med = lambda k: (round(st.median(r[k] for r in rows if r[k] is not None), 1) if any(r[k] is not None for r in rows) else None)
print(label, "iters", len(rows), " ".join(f"{k}={med(k)}" for k in ("iter", "side_n", "side_start", "side_end", "side_busy", "main_chain_busy",
                                                                       "main_chain_end", "comp_start", "comp_end")))
