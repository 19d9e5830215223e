#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1
for i in 1 2; do for k in keep drop; do
  python tools/exp/fwd_timeline_rv.py /tmp/r03 $k 2>/dev/null | tail -1
  python tools/exp/fwd_timeline_rv.py $GRAFT_REPO_ROOT $k 2>/dev/null | tail -1
done; done
