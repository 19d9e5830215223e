#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 > gpurun_out/r04_pytest_gpu_final.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04_pytest_gpu_final.log
bash tools/r04/final.sh
du -sh gpurun_out
