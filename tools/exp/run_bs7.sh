#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "depth_order or lists_without" 2>&1 | tail -3
GSR_DEPTH_SORT=bucket timeout 300 python tools/exp/bucket_sort_check.py 1000003 3000000 2>&1 | grep -v "amdgpu" | tail -28
for rep in 1 2; do
  timeout 300 python bench.py --no-pmc --no-cpu-baseline --train-iters 0 > gpurun_out/b.json 2>/dev/null
  python - gpurun_out/b.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v if isinstance(v, (int, float)) else v.get("ms", 0), 4) for n, v in (d.get("kernels") or {}).items()}
print(d["ms_per_step"], d["value"], {a: k[a] for a in ("depth_order", "bin_sorted")})
P
done
