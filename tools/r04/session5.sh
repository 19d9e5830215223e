#!/bin/bash
# round 4, GPU session 5: the whole GPU suite (per-test timeout), then the default bench with its train leg
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r04/pytest_gpu.log
(time python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err); echo "bench rc $?"; tail -3 gpurun_out/r04/bench_default.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_with_caller_syncs","caller_syncs_gap","ms_per_step_with_caller_and_camera_syncs")})
print("parity", json.dumps(d["parity_vs_oracle"])[:1500])
print("roofline", d["roofline"])
t=d["train"]
print("train", json.dumps(t)[:6000])
PY
