#!/bin/bash
# round 4, GPU session 8: exp2-staged compositing kernels -- timing, then the suite (short per-test timeout)
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 100 > gpurun_out/r04/bench_exp2.json 2> gpurun_out/r04/bench_exp2.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_exp2.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_median","ms_per_step_with_caller_syncs","caller_syncs_gap","ms_per_step_with_caller_and_camera_syncs")})
print({k:v["ms"] for k,v in d["kernels"].items()})
PY
python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 60 --scene longtail --no-synced-regions > gpurun_out/r04/bench_exp2_longtail.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_exp2_longtail.json"))
print("longtail", d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items() if k.startswith("raster")})
PY
timeout 2400 python -m pytest tests -m gpu -q --timeout 420 > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -30 gpurun_out/r04/pytest_gpu.log
