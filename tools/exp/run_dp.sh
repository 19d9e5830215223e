#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "several_views" 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -8
