#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "depth_order or lists_without" 2>&1 | tail -4
for m in "GSR_DEPTH_SORT=radix" "GSR_DEPTH_SORT=auto"; do
  env $m timeout 300 python bench.py --no-pmc --no-cpu-baseline --train-iters 0 --deterministic > gpurun_out/b.json 2>/dev/null
  python - gpurun_out/b.json "det $m" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v if isinstance(v, (int, float)) else v.get("ms", 0), 4) for n, v in (d.get("kernels") or {}).items()}
print(sys.argv[2], d["ms_per_step"], {a: k[a] for a in ("count_reach", "depth_order", "bin_sorted")})
P
  env $m timeout 300 python bench.py --no-pmc --no-cpu-baseline --train-iters 0 --scale-lo 0.005 --scale-hi 0.05 --gaussians 10000 --width 256 --height 256 --sh-degree 0 > gpurun_out/b.json 2>/dev/null
  python - gpurun_out/b.json "config1 $m" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v if isinstance(v, (int, float)) else v.get("ms", 0), 4) for n, v in (d.get("kernels") or {}).items()}
print(sys.argv[2], d["ms_per_step"], {a: k[a] for a in ("count_reach", "depth_order", "bin_sorted")})
P
done
