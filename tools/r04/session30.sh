#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python bench.py --train-only --train-iters 30000 > gpurun_out/r04/train_30k.json 2> gpurun_out/r04/train_30k.err; echo "rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/train_30k.json").read().strip().splitlines()[-1])
keep={k:d.get(k) for k in ("iters","iters_per_s","iters_per_s_with_caller_syncs","iters_per_s_unchanged_caller","psnr","refinements","list_overflow_views","schedule","phase_ms_median_by_resolution")}
keep["gaussians"]={k:v for k,v in d["gaussians"].items() if k!="history_step_N"}
for k in ("with_caller_syncs","full_resolution_from_step_0","unchanged_caller","refined_1m","fixed_1m","one_op_path"):
    v=d.get(k)
    if v: keep[k]={a:v.get(a) for a in ("iters_per_s","iters","gaussians_end","psnr","list_overflow_views")}
keep["command"]="python bench.py --train-only --train-iters 30000"
json.dump(keep,open("gpurun_out/r04/train_30k_summary.json","w"),indent=1)
print(json.dumps(keep)[:1500])
PY
