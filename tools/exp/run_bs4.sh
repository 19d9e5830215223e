#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for m in "GSR_DEPTH_SORT=radix GSR_FUSED_RECORDS=0" "GSR_FUSED_RECORDS=0" "GSR_FUSED_RECORDS=1"; do
  env $m timeout 300 python bench.py --no-pmc --no-cpu-baseline --train-iters 0 > gpurun_out/bench_bs.json 2>/dev/null
  python - gpurun_out/bench_bs.json "$m" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v if isinstance(v, (int, float)) else v.get("ms", 0), 4) for n, v in (d.get("kernels") or {}).items()}
print(sys.argv[2], d["ms_per_step"], d["value"], {a: k[a] for a in ("count_reach", "depth_order", "bin_sorted")})
P
done
done
