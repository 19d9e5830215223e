#!/bin/bash
# round 4, GPU session 3: lists built ahead of time on the side stream -- tests, A/B, traces
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_api.py -x -q --timeout 900 > gpurun_out/r04/pytest_api.log 2>&1; echo "pytest api rc $?"; tail -15 gpurun_out/r04/pytest_api.log
for sp in 0 sort lists; do
  GSR_SPECULATE=$sp python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 100 --event-every 0 > gpurun_out/r04/bench_spec_$sp.json 2> gpurun_out/r04/bench_spec_$sp.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_spec_$sp.json"))
print("speculate=$sp", {k:d[k] for k in ("value","ms_per_step","ms_per_step_with_caller_syncs","caller_syncs_gap","ms_per_step_with_caller_and_camera_syncs")})
PY
done
export TMPDIR=/tmp
for mode in off on camera; do
  rm -rf /tmp/prof_sync_$mode
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sync_$mode -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --train-iters 0 --event-every 0 --steps 20 --warmup 5 --caller-syncs $mode --no-synced-regions > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_sync_$mode.log 2>&1)
  python tools/step_seq.py /tmp/prof_sync_$mode gpurun_out/r04/step_sequence_ahead_caller_syncs_$mode.txt
done
cat gpurun_out/r04/step_sequence_ahead_caller_syncs_on.txt
