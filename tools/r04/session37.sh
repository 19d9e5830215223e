#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
BLOB=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seg -- python $GRAFT_REPO_ROOT/tools/exp/seg_ab.py 480 270 300000 30 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("/tmp/prof_seg/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))) for r in csv.DictReader(open(f))]
agg = collections.defaultdict(list)
for s, e, k, g in rows:
    if "raster_" in k:
        name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[(name[:44], g)].append((e - s) / 1e3)
for (k, g), v in sorted(agg.items()):
    v.sort()
    print(f"{k:46s} grid {g:8d} n={len(v):4d} median {v[len(v)//2]:8.1f} us")
PY
