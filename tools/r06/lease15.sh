#!/bin/bash
out=$PWD/gpurun_out/lease15; mkdir -p $out
( timeout 1700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 ) > $out/gpu_suite.txt; cat $out/gpu_suite.txt
