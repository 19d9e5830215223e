#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --timeout 420 -x -k "depth_segments or deep_tiles" 2>&1 | tail -15
for k in 8 16; do GSR_DEPTH_SEGMENTS=$k BLOB=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, runpy
sys.argv = ["seg_ab.py", "480", "270", "300000", "30"]
runpy.run_path("tools/exp/seg_ab.py", run_name="__main__")
PY
done 2>&1 | grep "segments 8\|segments 1:" | head -4
