#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
BLOB=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seg -- python $GRAFT_REPO_ROOT/tools/exp/seg_ab.py 480 270 300000 30 2>&1 | grep -v amdgpu.ids | tail -5
python - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/prof_seg/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))) for r in csv.DictReader(open(f))]
rows.sort()
import collections
agg = collections.defaultdict(list)
for s, e, k, g in rows:
    if "raster_bwd" in k:
        agg[(k.split("(")[0][-60:], g)].append((e - s) / 1e3)
for (k, g), v in agg.items():
    v.sort()
    print(f"{k:60s} grid {g:8d} n={len(v):4d} median {v[len(v)//2]:8.1f} us")
PY
cd $GRAFT_REPO_ROOT
for s in 1 4 8 1 8; do echo "GSR_DEPTH_SEGMENTS=$s"; GSR_DEPTH_SEGMENTS=$s python tools/exp/train_gpu_busy.py run 2>/dev/null | tail -1; done
