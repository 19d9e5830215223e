#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1
python tools/exp/gc_frames.py /tmp/r03 2>/dev/null | tail -3
python tools/exp/gc_frames.py $GRAFT_REPO_ROOT 2>/dev/null | tail -3
