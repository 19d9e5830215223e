#!/bin/bash
# round 4, GPU session 9: backward with the four pixel blocks of a splat fused into one basic block (GSR_BWD_FUSE4)
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-pmc --train-iters 0 --steps 80 --no-synced-regions"
for f in 0 1 0 1; do
  GSR_BWD_FUSE4=$f python bench.py $Q > gpurun_out/r04/fuse4_$f.json 2>/dev/null
  python - $f <<PY
import json, sys
d=json.load(open("gpurun_out/r04/fuse4_%s.json" % sys.argv[1]))
print("default fuse4", sys.argv[1], "ms", d["ms_per_step"], d["ms_per_step_median"], {k:v["ms"] for k,v in d["kernels"].items() if k.startswith("raster")})
PY
done
for f in 0 1; do
  GSR_BWD_FUSE4=$f python bench.py $Q --scene longtail > gpurun_out/r04/fuse4_lt_$f.json 2>/dev/null
  GSR_BWD_FUSE4=$f python bench.py $Q --scale-lo 0.005 --scale-hi 0.05 > gpurun_out/r04/fuse4_dense_$f.json 2>/dev/null
  python - $f <<PY
import json, sys
for nm in ("lt", "dense"):
    d=json.load(open("gpurun_out/r04/fuse4_%s_%s.json" % (nm, sys.argv[1])))
    print(nm, "fuse4", sys.argv[1], "ms", d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items() if k.startswith("raster")})
PY
done
GSR_BWD_FUSE4=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q --timeout 420 -x 2>&1 | tail -5
