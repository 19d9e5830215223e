#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
LEAN=1 python tools/exp/binbench.py 1000000 100 2>/dev/null | tail -1
LEAN=1 SORTED=1 python tools/exp/binbench.py 1000000 100 2>/dev/null | tail -1
done
