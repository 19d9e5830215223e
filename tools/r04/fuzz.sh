#!/bin/bash
# round 4: a randomised session over the fuzzers the round's changes touch, with the side stream forced on and in auto
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for sp in lists auto; do
  echo "== fuzz_sequence (400 calls, seed 4004, GSR_SPECULATE=$sp)"
  GSR_SPECULATE=$sp timeout 1500 python tools/exp/fuzz_sequence.py 400 4004 2>&1 | grep -v "amdgpu.ids" | tail -3
  echo "== fuzz_sequence, two-round lists forced (200 calls, seed 4005, GSR_SPECULATE=$sp)"
  GSR_SPECULATE=$sp GSR_TWO_ROUND=1 timeout 1500 python tools/exp/fuzz_sequence.py 200 4005 2>&1 | grep -v "amdgpu.ids" | tail -3
done
for f in lists raster render; do
  echo "== fuzz_$f (300 cases, seed 4006)"
  timeout 1500 python tools/exp/fuzz_$f.py 300 4006 2>&1 | grep -v "amdgpu.ids" | tail -2
done
} > gpurun_out/r04_fuzz.txt 2>&1
cat gpurun_out/r04_fuzz.txt
