#!/bin/bash
# what the driver does at round end, on a fresh box: the GPU suite, smoke(), then `python bench.py` under a clock
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/driver_like_pytest.log 2>&1); tail -3 gpurun_out/driver_like_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python bench.py > gpurun_out/driver_like_bench.json 2> gpurun_out/driver_like_bench.err); echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/driver_like_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","ms_per_step_median","value_with_caller_syncs","ms_per_step_with_caller_syncs","caller_syncs_gap","ms_per_step_with_caller_and_camera_syncs","higher_is_better","scaling","vs_baseline","dtype","data")})
print(d["roofline"]); print(d["cpu_baseline"]); print(d["parity_vs_oracle"]["meets"])
t=d["train"]; print({k:t.get(k) for k in ("iters_per_s","iters_per_s_with_caller_syncs","iters_per_s_unchanged_caller","wall_s_including_setup","list_overflow_views")})
PY
