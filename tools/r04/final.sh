#!/bin/bash
# round 4: everything DESIGN.md section 5 and profiles/ quote, from one GPU lease
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/collect_profiles.sh r04 > gpurun_out/collect_profiles_r04.log 2>&1
bash tools/collect_benches.sh r04 2>&1 | tee gpurun_out/collect_benches_r04.log
for sp in 0 lists auto; do for mode in off on camera; do
  GSR_SPECULATE=$sp python tools/exp/sync_timeline.py $mode 300 2>&1 | tail -1
done; done | tee gpurun_out/r04_sync_timeline.txt
python tools/render_bench.py > gpurun_out/r04_render_bench.txt 2>&1 || true
tail -8 gpurun_out/r04_render_bench.txt
