#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "depth_order" 2>&1 | tail -5
timeout 300 python tools/exp/bucket_sort_check.py 1000003 3000000 2>&1 | grep -v "equal\|two \|narrow\|tiny\|amdgpu" | tail -22
for m in radix auto; do
  GSR_DEPTH_SORT=$m timeout 300 python bench.py --no-pmc --no-cpu-baseline --train-iters 0 > gpurun_out/bench_bs_$m.json 2>/dev/null
  python - gpurun_out/bench_bs_$m.json $m <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v if isinstance(v, (int, float)) else v.get("ms", 0), 4) for n, v in (d.get("kernels") or {}).items()}
print(sys.argv[2], d["ms_per_step"], d["value"], k)
P
done
