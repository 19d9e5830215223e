#!/bin/bash
# round 4, GPU session 6: small tile grids (the coarse-to-fine schedule's first phases): every tile split over 4 waves
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-pmc --train-iters 0 --steps 60 --event-every 5 --no-synced-regions"
for sg in 0 2560; do
 for res in "480 270 300000" "960 540 450000"; do
  set -- $res
  GSR_SMALL_GRID=$sg python bench.py $Q --width $1 --height $2 --gaussians $3 --scale-lo 0.005 --scale-hi 0.03 > gpurun_out/r04/small_$1_$sg.json 2>/dev/null
  python - "$sg" "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r04/small_%s_%s.json" % (sys.argv[2], sys.argv[1])))
print("small_grid", sys.argv[1], sys.argv[2], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items() if k in ("raster_fwd", "raster_bwd", "bin_sorted", "depth_order")}, d["config"]["tile_list_length"])
PY
 done
done
for mn in 32 96 256; do
  GSR_SMALL_GRID_MIN=$mn python bench.py $Q --width 480 --height 270 --gaussians 300000 --scale-lo 0.005 --scale-hi 0.03 > gpurun_out/r04/small_min.json 2>/dev/null
  python - $mn <<PY
import json, sys
d=json.load(open("gpurun_out/r04/small_min.json"))
print("small_grid_min", sys.argv[1], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items() if k in ("raster_fwd", "raster_bwd")})
PY
done
for sg in 1024 4096 9000; do
  GSR_SMALL_GRID=$sg python bench.py $Q --width 960 --height 540 --gaussians 450000 --scale-lo 0.005 --scale-hi 0.03 > gpurun_out/r04/small_sg.json 2>/dev/null
  python - $sg <<PY
import json, sys
d=json.load(open("gpurun_out/r04/small_sg.json"))
print("960x540 small_grid", sys.argv[1], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items() if k in ("raster_fwd", "raster_bwd")})
PY
done
GSR_SMALL_GRID=9000 python bench.py $Q > gpurun_out/r04/small_1080.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r04/small_1080.json"))
print("1080p split-all ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items() if k in ("raster_fwd", "raster_bwd")})
PY
timeout 900 python -m pytest tests/test_gpu_nccl.py tests/test_gpu_bench.py -x -q --timeout 900 2>&1 | tail -15
