#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 1 4 8 1 4; do echo "bwd 8, fwd $f"; GSR_DEPTH_SEGMENTS_FWD=$f python tools/exp/train_gpu_busy.py run 2>/dev/null | tail -1; done
