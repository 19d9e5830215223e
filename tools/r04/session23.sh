#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_render.py tests/test_gpu_train.py -q --timeout 420 -x 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 100 --event-every 0 > gpurun_out/r04/prep.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r04/prep.json"))
print({k:d[k] for k in ("ms_per_step","ms_per_step_median","ms_per_step_with_caller_syncs","caller_syncs_gap","ms_per_step_with_caller_and_camera_syncs")})
PY
done
GSR_SPECULATE=auto python tools/exp/sync_timeline.py on 300 2>&1 | tail -1 | cut -c1-260
