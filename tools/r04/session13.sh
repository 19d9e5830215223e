#!/bin/bash
# round 4, GPU session 13: GSR_SPECULATE = 0 / lists / auto on ONE box: unsynced, synced, camera; three scenes; forward only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
Q="--no-cpu-baseline --no-pmc --train-iters 0 --steps 100 --event-every 0"
run() {  # label, extra flags
  for sp in 0 lists auto; do
    GSR_SPECULATE=$sp python bench.py $Q $2 > gpurun_out/r04/spec_ab.json 2>/dev/null
    python - "$1" $sp <<PY
import json, sys
d=json.load(open("gpurun_out/r04/spec_ab.json"))
print("%-10s speculate=%-5s unsynced %.4f  synced %.4f  camera %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["ms_per_step_with_caller_syncs"], d["ms_per_step_with_caller_and_camera_syncs"]))
PY
  done
}
run default ""
run config2 "--scale-lo 0.005 --scale-hi 0.05 --gaussians 200000"
run dense "--scale-lo 0.005 --scale-hi 0.05"
run res540 "--width 960 --height 540 --gaussians 450000 --scale-lo 0.005 --scale-hi 0.03"
for sp in 0 lists auto; do
  echo "forward only speculate=$sp: $(GSR_SPECULATE=$sp python tools/render_bench.py 2>/dev/null | tail -1 | cut -c1-140)"
done
