#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" python tools/exp/train_gpu_busy.py run 2>/dev/null | tail -1; }
run GSR_P2_MIN=1000000
run GSR_P2_MIN=1000000
run GSR_P2_MIN=300000
run GSR_P2_MIN=1000000
run GSR_P2_MIN=100000
run GSR_P2_MIN=1000000
run GSR_P2_MIN=600000
