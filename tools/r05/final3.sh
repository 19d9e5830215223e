#!/bin/bash
# The suite and the driver's form of the bench on the final code.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > "$O/gpu_suite.log" 2>&1
tail -18 "$O/gpu_suite.log"
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 2> "$O/final3_$i.err" | tail -1 > "$O/bench_r05_driver_form_$i.json"
  python tools/r05/digest.py "$O/bench_r05_driver_form_$i.json"
done
timeout 600 python bench.py 2> "$O/final3_default.err" | tail -1 > "$O/bench_r05_default.json"
python tools/r05/digest.py "$O/bench_r05_default.json"
