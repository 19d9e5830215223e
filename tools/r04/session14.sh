#!/bin/bash
cd $GRAFT_REPO_ROOT
for envs in "GSR_SPECULATE=0" "GSR_SPECULATE=0 GSR_ONE_CALL=0" "GSR_SPECULATE=0 GSR_POLL_YIELD=0" "GSR_SPECULATE=0 GSR_ONE_CALL=0 GSR_POLL_YIELD=0" "GSR_SPECULATE=0 GSR_NO_SPECULATION=1"; do
  echo "$envs: $(env $envs python tools/render_bench.py 2>/dev/null | tail -1 | cut -c1-120)"
done
python tools/exp/host_prof_fwd.py 2>&1 | head -40
