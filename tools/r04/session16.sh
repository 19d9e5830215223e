#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1
for i in 1 2; do
  python tools/exp/fwd_timeline.py /tmp/r03 2>/dev/null | tail -1
  python tools/exp/fwd_timeline.py $GRAFT_REPO_ROOT 2>/dev/null | tail -1
  GSR_SPECULATE=0 GSR_ONE_CALL=0 python tools/exp/fwd_timeline.py $GRAFT_REPO_ROOT 2>/dev/null | tail -1
done
