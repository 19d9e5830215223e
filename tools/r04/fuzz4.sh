#!/bin/bash
# the final state of round 4 under every fuzzer, fresh seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=200; SEED=4020
for f in lists depth_order project raster sequence render refine fused; do
  echo "== fuzz_$f ($C cases, seed $SEED)"
  timeout 1500 python tools/exp/fuzz_$f.py $C $SEED 2>&1 | grep -v "amdgpu.ids" | tail -3
done > gpurun_out/r04_fuzz4.txt 2>&1
grep -c " ok" gpurun_out/r04_fuzz4.txt; grep "==\|mismatches" gpurun_out/r04_fuzz4.txt
