#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" python tools/exp/train_gpu_busy.py run 2>/dev/null | tail -1; }
run GSR_DEPTH_SEGMENTS=16
run GSR_DEPTH_SEGMENTS=16
run GSR_DEPTH_SEGMENTS=32
run GSR_DEPTH_SEGMENTS=64
run GSR_DEPTH_SEGMENTS=16
run GSR_DEPTH_SEGMENTS=32
run GSR_DEPTH_SEGMENTS=64 GSR_DEPTH_SEGMENTS_MIN=1024
