#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/exp/host_prof.py 300000 480 270 3 2>&1 | head -60
