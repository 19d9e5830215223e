#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1; tail -1 /tmp/r03/build.log
for i in 1 2 3; do
  echo "r03 code      : $(cd /tmp/r03 && python tools/render_bench.py 2>/dev/null | tail -1 | cut -c50-110)"
  echo "now (auto)    : $(python tools/render_bench.py 2>/dev/null | tail -1 | cut -c50-110)"
  echo "now ONE_CALL=0: $(GSR_ONE_CALL=0 python tools/render_bench.py 2>/dev/null | tail -1 | cut -c50-110)"
  echo "now SPEC=0    : $(GSR_SPECULATE=0 python tools/render_bench.py 2>/dev/null | tail -1 | cut -c50-110)"
done
