#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04

timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python tools/exp/train_gpu_busy.py run > /tmp/prof_run.log 2>&1; tail -1 /tmp/prof_run.log
python tools/exp/train_gpu_busy.py summarize /tmp/prof_train | tee gpurun_out/r04/train_gpu_busy.txt
