#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/exp/op_host_time.py 300000 480 270 2>&1 | tail -30
