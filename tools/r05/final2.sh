#!/bin/bash
# The closing records on the final code: the whole GPU suite, the rocprofv3 evidence for profiles/ (kernel trace,
# step sequences, separate --pmc passes), the driver's form of the bench, the host timeline with the models' read-backs.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
rm -f "$O/gpu_suite.log"
timeout 1200 python -m pytest tests -q -m gpu > "$O/gpu_suite.log" 2>&1
tail -3 "$O/gpu_suite.log"
timeout 900 bash tools/collect_profiles.sh r05 > "$O/collect_r05.log" 2>&1
cd "$R"
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 2> "$O/final2_$i.err" | tail -1 > "$O/bench_r05_driver_form_$i.json"
  python tools/r05/digest.py "$O/bench_r05_driver_form_$i.json"
done
for M in on camera off; do
  timeout 200 python tools/exp/sync_timeline.py $M 300 2>/dev/null | tail -1
done
