#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py -q --timeout 300 -x -k "depth_segments or render_gaussians_equals" 2>&1 | tail -2
for f in 8 16 8; do echo "forward runs $f"; GSR_DEPTH_SEGMENTS_FWD=$f python tools/exp/train_gpu_busy.py run 2>/dev/null | tail -1; done
BLOB=1 SEGS=16 python tools/exp/seg_ab.py 480 270 300000 30 2>&1 | grep "^segments" | cut -c1-100
