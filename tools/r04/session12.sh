#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
Q="--no-cpu-baseline --no-pmc --train-iters 0 --steps 20 --warmup 5 --no-synced-regions --scale-lo 0.005 --scale-hi 0.05 --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth"
for envs in "GSR_SPECULATE=lists" "GSR_SPECULATE=0" "GSR_SPECULATE=sort" "GSR_SPECULATE=0 GSR_ONE_CALL=0"; do
  env $envs python bench.py $Q > gpurun_out/r04/c5.json 2> gpurun_out/r04/c5.err
  python - "$envs" <<PY
import json, sys
d=json.load(open("gpurun_out/r04/c5.json"))
print(sys.argv[1], "ms", d["ms_per_step"], d["ms_per_step_median"], d["config"]["two_round_lists"], {k:(v["ms"], v["calls_per_step"]) for k,v in d["kernels"].items() if v["ms"]})
PY
done
