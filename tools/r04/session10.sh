#!/bin/bash
# round 4, GPU session 10: the backward's fused four-pixel block against the kernel as committed, on ONE box
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-pmc --train-iters 0 --steps 80 --no-synced-regions"
bench() {  # label, env
  env $2 python bench.py $Q > gpurun_out/r04/ab.json 2>/dev/null
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r04/ab.json"))
print(sys.argv[1], "ms", d["ms_per_step"], d["ms_per_step_median"], {k:v["ms"] for k,v in d["kernels"].items() if k.startswith("raster")})
PY
}
bench "macro fuse4=0" GSR_BWD_FUSE4=0
bench "macro fuse4=1" GSR_BWD_FUSE4=1
cp gaussian-splatting-toolkit_amd/csrc/raster_bwd.hip /tmp/raster_bwd_new.hip
cp tools/r04/raster_bwd_head.hip.txt gaussian-splatting-toolkit_amd/csrc/raster_bwd.hip
make -C gaussian-splatting-toolkit_amd/csrc -j8 > /dev/null 2>&1
bench "committed kernel" GSR_BWD_FUSE4=0
bench "committed kernel" GSR_BWD_FUSE4=0
cp /tmp/raster_bwd_new.hip gaussian-splatting-toolkit_amd/csrc/raster_bwd.hip
make -C gaussian-splatting-toolkit_amd/csrc -j8 > /dev/null 2>&1
bench "macro fuse4=1" GSR_BWD_FUSE4=1
bench "macro fuse4=0" GSR_BWD_FUSE4=0
