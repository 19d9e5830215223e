#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 > gpurun_out/r04_pytest_gpu_final.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04_pytest_gpu_final.log
python bench.py --no-cpu-baseline --no-pmc --train-iters 7000 --steps 100 > gpurun_out/r04_bench_quick.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r04_bench_quick.json"))
print({k:d[k] for k in ("ms_per_step","ms_per_step_median","ms_per_step_with_caller_syncs","ms_per_step_with_caller_and_camera_syncs")})
t=d["train"]; print({k:t.get(k) for k in ("iters_per_s","iters_per_s_with_caller_syncs","iters_per_s_unchanged_caller")}, t["phase_ms_median_by_resolution"])
PY
