#!/usr/bin/env python3
"""Idle time between consecutive kernels of the steady state, from a rocprofv3 --kernel-trace directory:
why does a HIP-graph replay of a view lose to eager launches?   python tools/exp/gap_stats.py <dir>"""
import csv, glob, json, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]  # steady state
st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
gap = (st[1:] - en[:-1]) / 1e3
busy = (en - st).sum() / 1e3
span = (en[-1] - st[0]) / 1e3
iters = sum("project_fwd" in r["Kernel_Name"] for r in rows)
print(json.dumps({"kernels": len(rows), "iterations": iters, "kernels_per_iteration": round(len(rows) / max(iters, 1), 1),
                  "span_us_per_iteration": round(span / max(iters, 1), 1), "busy_us_per_iteration": round(busy / max(iters, 1), 1),
                  "gap_us_median": round(float(np.median(gap)), 2), "gap_us_mean": round(float(gap.clip(min=0).mean()), 2),
                  "gap_us_p90": round(float(np.percentile(gap, 90)), 2),
                  "idle_us_per_iteration": round(float(gap.clip(min=0).sum()) / max(iters, 1), 1)}))
