#!/bin/bash
# round 4, GPU session 2: where the caller-sync gap sits (kernel trace of a synced run), what the read-backs cost
# under different runtime environments, then the GPU suite with a per-test timeout
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
for envs in "" "HSA_ENABLE_SDMA=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_SDMA=0 HIP_FORCE_DEV_KERNARG=1"; do
  env $envs python tools/exp/sync_latency.py >> gpurun_out/r04/sync_latency.jsonl 2>> gpurun_out/r04/sync_latency.err
done
cat gpurun_out/r04/sync_latency.jsonl
for envs in "HSA_ENABLE_SDMA=0" "HIP_FORCE_DEV_KERNARG=1"; do
  env $envs python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 100 --event-every 0 > gpurun_out/r04/bench_env.json 2>> gpurun_out/r04/bench_env.err
  python - "$envs" <<PY
import json, sys
d=json.load(open("gpurun_out/r04/bench_env.json"))
print(sys.argv[1], {k:d[k] for k in ("ms_per_step","ms_per_step_with_caller_syncs","ms_per_step_with_caller_and_camera_syncs")})
PY
done
export TMPDIR=/tmp
for mode in on camera; do
  rm -rf /tmp/prof_sync_$mode
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sync_$mode -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --train-iters 0 --event-every 0 --steps 20 --warmup 5 --caller-syncs $mode > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_sync_$mode.log 2>&1)
  python tools/step_seq.py /tmp/prof_sync_$mode gpurun_out/r04/step_sequence_caller_syncs_$mode.txt
done
cat gpurun_out/r04/step_sequence_caller_syncs_on.txt
timeout 2000 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04/pytest_gpu.log
