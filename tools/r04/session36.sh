#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py -q --timeout 420 -x -k "depth_segments or deep_tiles or render_gaussians_equals" 2>&1 | tail -15
BLOB=1 timeout 300 python tools/exp/seg_ab.py 480 270 300000 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/exp/seg_ab.py 480 270 300000 2>&1 | grep -v amdgpu.ids
