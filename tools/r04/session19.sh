#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1
export TMPDIR=/tmp
for v in r03 now; do
  root=$GRAFT_REPO_ROOT; [ $v = r03 ] && root=/tmp/r03
  rm -rf /tmp/prof_$v
  (cd /tmp && GSR_SPECULATE=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -- python $root/tools/render_bench.py --frames 60 > /tmp/prof_$v.log 2>&1)
  tail -1 /tmp/prof_$v.log | cut -c50-110
  python - /tmp/prof_$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "project_fwd" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]   # one full frame in steady state
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
print(sys.argv[2], "frame period us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]))
    prev = e
PY
done
