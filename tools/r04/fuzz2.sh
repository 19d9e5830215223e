#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== fuzz_sequence (400 calls, seed 4007, GSR_SPECULATE=lists GSR_SPECULATE_MIN=1: lists built ahead at every size)"
GSR_SPECULATE=lists GSR_SPECULATE_MIN=1 timeout 1500 python tools/exp/fuzz_sequence.py 400 4007 2>&1 | grep -v "amdgpu.ids" | tail -3
echo "== fuzz_sequence, two-round lists forced (200 calls, seed 4008, GSR_SPECULATE=lists GSR_SPECULATE_MIN=1)"
GSR_SPECULATE=lists GSR_SPECULATE_MIN=1 GSR_TWO_ROUND=1 timeout 1500 python tools/exp/fuzz_sequence.py 200 4008 2>&1 | grep -v "amdgpu.ids" | tail -3
} > gpurun_out/r04_fuzz2.txt 2>&1
cat gpurun_out/r04_fuzz2.txt
