#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/exp/seg_ab.py 480 270 300000 2>&1 | grep -v amdgpu.ids
BLOB=1 timeout 300 python tools/exp/seg_ab.py 480 270 300000 2>&1 | grep -v amdgpu.ids
BLOB=1 timeout 300 python tools/exp/seg_ab.py 960 540 450000 2>&1 | grep -v amdgpu.ids
