#!/bin/bash
cd $GRAFT_REPO_ROOT
D=gaussian-splatting-toolkit_amd/rasterizer/cuda
echo "== new"; python tools/exp/project_ab.py; python tools/exp/project_ab.py
cp $D/libgsraster.so /tmp/new.so; cp $D/libgsraster_base.so $D/libgsraster.so
echo "== base"; python tools/exp/project_ab.py; python tools/exp/project_ab.py
cp /tmp/new.so $D/libgsraster.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -q --timeout 420 -x 2>&1 | tail -3
