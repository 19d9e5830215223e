#!/bin/bash
out=$PWD/gpurun_out/lease10; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_heldout.py -q -k needles -s 2>&1 | grep -E "max \|err\||L2 relative|stable Gaussians|worst unstable|Error|passed|failed" | cut -c1-260 > $out/needles.txt; cat $out/needles.txt
tools/exp/dispatch_bench 2>&1 | tail -6 | tee $out/dispatch_regs.txt
