#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_dp.py tests/test_gpu_nccl.py -q --timeout 420 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 100 --event-every 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('ms_per_step','ms_per_step_median','ms_per_step_with_caller_syncs','caller_syncs_gap','ms_per_step_with_caller_and_camera_syncs')})"
