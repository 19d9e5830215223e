#!/bin/bash
cd $GRAFT_REPO_ROOT
export W=3840 H=2160 SCALE_LO=0.005 SCALE_HI=0.05 LEAN=1
python tools/exp/binbench.py 3000000 20 2>/dev/null | tail -1
SORTED=1 python tools/exp/binbench.py 3000000 20 2>/dev/null | tail -1
