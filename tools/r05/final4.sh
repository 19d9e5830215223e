#!/bin/bash
# Closing records on the final code (staged-ahead loads in): the whole GPU suite, the rocprofv3 evidence of
# tools/collect_profiles.sh r05, the driver's form of the bench twice, the default bench, config 5 / long-tail / config 2.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > "$O/gpu_suite.log" 2>&1
tail -18 "$O/gpu_suite.log"
timeout 900 bash tools/collect_profiles.sh r05 > "$O/collect_r05.log" 2>&1
cd "$R"
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$O/final4_$i.err" | tail -1 > "$O/bench_r05_driver_form_$i.json"
  python tools/r05/digest.py "$O/bench_r05_driver_form_$i.json"
done
timeout 600 python bench.py 2> "$O/final4_default.err" | tail -1 > "$O/bench_r05_default.json"
python tools/r05/digest.py "$O/bench_r05_default.json"
timeout 300 python bench.py --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --train-iters 0 --no-cpu-baseline 2>/dev/null | tail -1 > "$O/bench_r05_config5_fused_depth.json"
timeout 300 python bench.py --scene longtail --train-iters 0 --no-cpu-baseline 2>/dev/null | tail -1 > "$O/bench_r05_longtail.json"
timeout 300 python bench.py --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05 --train-iters 0 2>/dev/null | tail -1 > "$O/bench_r05_config2.json"
python tools/r05/digest.py "$O/bench_r05_config5_fused_depth.json" "$O/bench_r05_longtail.json" "$O/bench_r05_config2.json"
