#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1
for i in 1 2 3; do
  echo "r03 code          : $(cd /tmp/r03 && python tools/render_bench.py 2>/dev/null | tail -1 | cut -c64-100)"
  echo "now               : $(python tools/render_bench.py 2>/dev/null | tail -1 | cut -c64-100)"
  echo "now POLL_YIELD=0  : $(GSR_POLL_YIELD=0 python tools/render_bench.py 2>/dev/null | tail -1 | cut -c64-100)"
  echo "now SPEC=0 YIELD=0: $(GSR_SPECULATE=0 GSR_POLL_YIELD=0 python tools/render_bench.py 2>/dev/null | tail -1 | cut -c64-100)"
done
