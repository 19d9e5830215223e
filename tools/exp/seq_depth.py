"""Kernel sequence of one depth_order call from a rocprofv3 --kernel-trace csv dir: python tools/exp/seq_depth.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "hist_kernel<true>" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + int(sys.argv[2]) if len(sys.argv) > 2 else i0 + 16]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f  %s" % ((a - t0) / 1000, (b - a) / 1000, r["Kernel_Name"][:80]))
