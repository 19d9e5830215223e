#!/bin/bash
# round 4, GPU session 1: caller-sync gap with and without the one-call forward; then the GPU suite
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
for oc in 0 1; do
  GSR_ONE_CALL=$oc python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 100 --event-every 0 > gpurun_out/r04/bench_onecall$oc.json 2> gpurun_out/r04/bench_onecall$oc.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_onecall$oc.json"))
print("one_call=$oc", {k:d[k] for k in ("value","ms_per_step","ms_per_step_with_caller_syncs","caller_syncs_gap","ms_per_step_with_caller_and_camera_syncs")})
PY
done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04/pytest_gpu.log
