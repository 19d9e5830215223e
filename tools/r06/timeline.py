#!/usr/bin/env python3
"""Merged GPU / host timeline from a rocprofv3 --kernel-trace --hip-trace CSV directory.
    python tools/r06/timeline.py <dir> <out.txt> [<window.csv.gz>]
Iterations are cut at project_fwd launches.  Over the second half of the run: per-iteration GPU busy / idle time,
kernels per queue / stream, where the idle gaps sit (which kernel follows them), per-HIP-call host time; then the
last 4 iterations as a merged listing (kernels with queue id, HIP calls of >= 5 us and every synchronising call)."""
import collections
import csv
import glob
import gzip
import sys

d, out_txt = sys.argv[1], sys.argv[2]
win_csv = sys.argv[3] if len(sys.argv) > 3 else None
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
hf = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)
K = list(csv.DictReader(open(kf[0]))) if kf else []
H = list(csv.DictReader(open(hf[0]))) if hf else []
o = open(out_txt, "w")
o.write("kernel columns: %s\nhip columns: %s\n" % (list(K[0].keys()) if K else None, list(H[0].keys()) if H else None))
for r in K:
    r["a"], r["b"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
for r in H:
    r["a"], r["b"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
K.sort(key=lambda r: r["a"])
H.sort(key=lambda r: r["a"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
qkey = lambda r: (r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
cuts = [r["a"] for r in K if "project_fwd" in r["Kernel_Name"]]
o.write("kernels %d, hip calls %d, iterations %d\n" % (len(K), len(H), len(cuts)))
if len(cuts) < 20:
    o.write("too few iterations\n")
    sys.exit(0)
first = len(cuts) // 2
# ---- per iteration: busy, idle, span
stats = []
gap_after = collections.Counter()
gap_us = collections.defaultdict(float)
queues = collections.Counter()
ki = 0
for it in range(first, len(cuts) - 1):
    t0, t1 = cuts[it], cuts[it + 1]
    rows = [r for r in K if t0 <= r["a"] < t1]
    busy = 0
    end = t0
    idle = 0
    for r in rows:
        queues[qkey(r)] += 1
        if r["a"] > end:
            g = (r["a"] - end) / 1e3
            idle += g
            if g >= 3.0:
                gap_after[short(r["Kernel_Name"])] += 1
                gap_us[short(r["Kernel_Name"])] += g
        end = max(end, r["b"])
    # union busy
    busy = (end - t0) / 1e3 - idle
    stats.append(((t1 - t0) / 1e3, busy, idle, len(rows)))
import statistics as st

sp = [s[0] for s in stats]
o.write("\nper iteration over %d iterations (us): span p10 %.1f p50 %.1f p90 %.1f | busy p50 %.1f | idle p50 %.1f p90 %.1f | kernels p50 %d\n" % (
    len(stats), sorted(sp)[len(sp) // 10], st.median(sp), sorted(sp)[9 * len(sp) // 10], st.median(s[1] for s in stats),
    st.median(s[2] for s in stats), sorted(s[2] for s in stats)[9 * len(stats) // 10], st.median(s[3] for s in stats)))
o.write("kernels per (queue, stream): %s\n" % dict(queues))
o.write("\nidle gaps >= 3 us by the kernel that FOLLOWS them (count, mean us, us per iteration):\n")
for k, c in gap_after.most_common(25):
    o.write("  %-60s %6d %8.1f %8.1f\n" % (k, c, gap_us[k] / c, gap_us[k] / len(stats)))
# ---- kernel time per iteration by name
kt = collections.defaultdict(float)
kn = collections.Counter()
for r in K:
    if cuts[first] <= r["a"] < cuts[-1]:
        kt[short(r["Kernel_Name"])] += (r["b"] - r["a"]) / 1e3
        kn[short(r["Kernel_Name"])] += 1
o.write("\nkernel time per iteration (us), launches per iteration:\n")
n_it = len(cuts) - 1 - first
for k, v in sorted(kt.items(), key=lambda x: -x[1])[:40]:
    o.write("  %-60s %8.1f %6.2f\n" % (k, v / n_it, kn[k] / n_it))
# ---- host: HIP calls per iteration
ht = collections.defaultdict(float)
hn = collections.Counter()
hmax = collections.defaultdict(float)
for r in H:
    if cuts[first] <= r["a"] < cuts[-1]:
        ht[r["Function"]] += (r["b"] - r["a"]) / 1e3
        hn[r["Function"]] += 1
        hmax[r["Function"]] = max(hmax[r["Function"]], (r["b"] - r["a"]) / 1e3)
o.write("\nHIP API host time per iteration (us), calls per iteration, longest call (us):\n")
for k, v in sorted(ht.items(), key=lambda x: -x[1])[:30]:
    o.write("  %-40s %8.1f %6.2f %9.1f\n" % (k, v / n_it, hn[k] / n_it, hmax[k]))
threads = collections.Counter(r.get("Thread_Id") for r in H)
o.write("HIP calls per thread: %s\n" % dict(threads))
# ---- merged listing of the last 4 iterations
w0, w1 = cuts[-6], cuts[-2]
ev = []
for r in K:
    if w0 <= r["a"] < w1:
        ev.append((r["a"], "GPU q%s s%s" % qkey(r), short(r["Kernel_Name"]), (r["b"] - r["a"]) / 1e3))
syncish = ("Synchronize", "Query", "WaitEvent", "Memcpy", "EventRecord", "Malloc", "Free")
for r in H:
    if w0 - 2_000_000 <= r["a"] < w1:
        dur = (r["b"] - r["a"]) / 1e3
        if True:
            ev.append((r["a"], "host t%s" % r.get("Thread_Id", "?")[-4:], r["Function"], dur))
ev.sort()
o.write("\nmerged listing, last 4 iterations (t us from the window start; dur us):\n")
for a, who, name, dur in ev:
    if a >= w0 - 300_000:
        o.write("%10.1f  %-16s %9.1f  %s\n" % ((a - w0) / 1e3, who, dur, name))
o.close()
if win_csv:
    with gzip.open(win_csv, "wt") as g:
        g.write("kind,who,name,start_ns,end_ns,corr\n")
        for r in K:
            if cuts[-40] <= r["a"] < cuts[-1]:
                g.write("k,q%s/s%s,%s,%d,%d,%s\n" % (qkey(r) + (short(r["Kernel_Name"]).replace(",", ";"), r["a"], r["b"], r.get("Correlation_Id", ""))))
        for r in H:
            if cuts[-40] <= r["a"] < cuts[-1]:
                g.write("h,t%s,%s,%d,%d,%s\n" % (r.get("Thread_Id", "?"), r["Function"], r["a"], r["b"], r.get("Correlation_Id", "")))
