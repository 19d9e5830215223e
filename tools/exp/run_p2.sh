#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $ROOT/bench.py --no-cpu-baseline --no-pmc --train-iters 0 --event-every 0 --steps 25 --warmup 5 > /tmp/prof_kt.log 2>&1
python $ROOT/tools/step_seq.py /tmp/prof_kt /tmp/seq.txt; grep -n "gsr_p2\|gsr_bsort" /tmp/seq.txt | cut -c1-100
