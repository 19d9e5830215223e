#!/usr/bin/env python3
"""Kernel sequence of the LAST bench step from a rocprofv3 --kernel-trace CSV directory.
    python tools/step_seq.py <dir> <out.txt>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "project_fwd" in r["Kernel_Name"]]
s = idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
prev_end = t0
with open(sys.argv[2], "w") as out:
    out.write("start_us   dur_us   gap_us  kernel\n")
    for r in rows[s:]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        out.write("%8.1f %8.1f %8.1f  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, name))
        prev_end = b
