#!/bin/bash
# round 4, after the depth segments: the fuzzers that reach the compositing kernels, with the segments at their defaults
# and forced onto short lists (minimum 64 entries, an odd number of runs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== fuzz_raster (300 cases, seed 4010, defaults: 16 runs above 512 entries on grids of up to 1100 tiles)"
timeout 1500 python tools/exp/fuzz_raster.py 300 4010 2>&1 | grep -v "amdgpu.ids" | tail -2
echo "== fuzz_raster (300 cases, seed 4011, GSR_DEPTH_SEGMENTS=5 GSR_DEPTH_SEGMENTS_MIN=64)"
GSR_DEPTH_SEGMENTS=5 GSR_DEPTH_SEGMENTS_MIN=64 timeout 1500 python tools/exp/fuzz_raster.py 300 4011 2>&1 | grep -v "amdgpu.ids" | tail -2
echo "== fuzz_render (200 cases, seed 4012, GSR_DEPTH_SEGMENTS_MIN=64)"
GSR_DEPTH_SEGMENTS_MIN=64 timeout 1500 python tools/exp/fuzz_render.py 200 4012 2>&1 | grep -v "amdgpu.ids" | tail -2
echo "== fuzz_fused (200 cases, seed 4013, GSR_DEPTH_SEGMENTS_MIN=64)"
GSR_DEPTH_SEGMENTS_MIN=64 timeout 1500 python tools/exp/fuzz_fused.py 200 4013 2>&1 | grep -v "amdgpu.ids" | tail -2
echo "== fuzz_sequence (300 calls, seed 4014, GSR_SPECULATE=lists GSR_DEPTH_SEGMENTS_MIN=64)"
GSR_SPECULATE=lists GSR_DEPTH_SEGMENTS_MIN=64 timeout 1500 python tools/exp/fuzz_sequence.py 300 4014 2>&1 | grep -v "amdgpu.ids" | tail -3
} > gpurun_out/r04_fuzz3.txt 2>&1
cat gpurun_out/r04_fuzz3.txt
