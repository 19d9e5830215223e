#!/bin/bash
# round 4, GPU session 7: the whole GPU suite without -x (all failures at once)
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -40 gpurun_out/r04/pytest_gpu.log
