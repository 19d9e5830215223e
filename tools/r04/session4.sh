#!/bin/bash
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
for sp in 0 lists; do for mode in off on camera; do
  GSR_SPECULATE=$sp python tools/exp/sync_timeline.py $mode 300 2>&1 | tail -1
done; done | tee gpurun_out/r04/sync_timeline.txt
