#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -5
for m in views dense; do
timeout 600 python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --no-pmc --no-cpu-baseline --train-iters 0 --sh-exchange $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['allreduce_bytes'], d['config']['parallelism'][:90])"
done
