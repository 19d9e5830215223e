#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_segpmc
BLOB=1 SEGS=16 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_segpmc -- python $GRAFT_REPO_ROOT/tools/exp/seg_ab.py 480 270 300000 6 > /tmp/segpmc.log 2>&1
tail -2 /tmp/segpmc.log
ls /tmp/prof_segpmc/*/ | head
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("/tmp/prof_segpmc/**/*counter_collection.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "raster_" not in k:
        continue
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    agg[(name, g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, g), c in sorted(agg.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0)
    simd_cycles = gui / 8 * 1024 if gui else 0
    busy = 4 * m.get("SQ_INSTS_VALU", 0) / simd_cycles if simd_cycles else 0
    act = 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles if simd_cycles else 0
    print(f"{name:46s} grid {g:>9s} n={len(c.get('SQ_WAVES', []))} waves {m.get('SQ_WAVES', 0):9.0f} VALU insts {m.get('SQ_INSTS_VALU', 0) / 1e6:7.2f} M  GUI_ACTIVE {gui / 1e6:6.3f} M  valu_busy {busy:.3f}")
PY
