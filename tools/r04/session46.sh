#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py -q --timeout 300 -x -k "depth_segments or render_gaussians_equals" 2>&1 | tail -3
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_seg
BLOB=1 SEGS=8,16 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seg -- python $GRAFT_REPO_ROOT/tools/exp/seg_ab.py 480 270 300000 30 2>&1 | grep "^segments" | cut -c1-110
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
cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/exp/train_gpu_busy.py run 2>/dev/null | tail -1; done
