#!/usr/bin/env python3
"""Reduce rocprofv3 output directories to small per-kernel summaries.

    python tools/summarize_prof.py <dir> <out.json>

Reads every *kernel_trace.csv / *counter_collection.csv / *kernel_stats.csv
under <dir> and writes per-kernel launch counts, mean/min/max duration (ns) and
mean counter values per dispatch."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\s*\[clone .*\]$", "", name)
    m = re.match(r"(?:void\s+)?([A-Za-z0-9_:<>]+?)(?:<.*)?\(", name)
    base = m.group(1) if m else name
    if "rocprim" in name:
        k = re.search(r"rocprim::(?:ROCPRIM_\w+::)?detail::(\w+)", name)
        base = "rocprim::" + (k.group(1) if k else "kernel")
    return base[:80]


def main(d, out):
    res = {"kernels": defaultdict(lambda: {"n": 0, "dur_ns": []}), "counters": defaultdict(lambda: defaultdict(list))}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            res["kernels"][k]["n"] += 1
            res["kernels"][k]["dur_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            res["counters"][k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if len(sys.argv) > 3:
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        prev_end = None
        for r in rows[-int(sys.argv[3]):]:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (st - prev_end) / 1e3 if prev_end else 0.0
            prev_end = en
            grid = r.get("Grid_Size") or r.get("Grid_Size_X", "?")
            wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X", "?")
            print(f"  {short(r['Kernel_Name']):50s} grid={grid:>9s} wg={wg:>4s} "
                  f"vgpr={r.get('VGPR_Count', '?'):>4s} dur={(en - st) / 1e3:8.1f}us gap={gap:7.1f}us")
    summ = {"kernels": {}, "counters": {}}
    for k, v in res["kernels"].items():
        ds = v["dur_ns"]
        summ["kernels"][k] = {"launches": v["n"], "mean_ns": sum(ds) / len(ds), "min_ns": min(ds), "max_ns": max(ds),
                              "total_ms": sum(ds) / 1e6}
    for k, cs in res["counters"].items():
        summ["counters"][k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
    json.dump(summ, open(out, "w"), indent=1, sort_keys=True)
    tot = sum(v["total_ms"] for v in summ["kernels"].values())
    for k, v in sorted(summ["kernels"].items(), key=lambda kv: -kv[1]["total_ms"])[:25]:
        print(f"{k:60s} n={v['launches']:5d} mean={v['mean_ns'] / 1e3:9.1f}us  {100 * v['total_ms'] / max(tot, 1e-9):5.1f}%")
    for k, cs in summ["counters"].items():
        print(k, {c: round(v["mean"], 1) for c, v in cs.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
