#!/bin/bash
# Why does the 1 M "ball" training leg (bench.py fixed_1m) lose 7 % under the ordered launch?
out=${1:-gpurun_out/ball}; mkdir -p $out
run() {
  local label=$1; shift
  python tools/train_bench.py --iters 400 --sh-interval 100 --phase-every 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$label', 'it/s', round(d['iters_per_s'], 1), 'phases', d.get('phase_ms_median'))"
}
for rep in 1 2; do
GSR_DEEP_ORDER=0 run "static"
GSR_DEEP_ORDER=1 run "ordered tail=8 bwdfac=2.0"
GSR_DEEP_ORDER=1 GSR_DEEP_TAIL=0 run "ordered tail=0 bwdfac=2.0"
GSR_DEEP_ORDER=1 GSR_DEEP_FACTOR_BWD=1.2 run "ordered tail=8 bwdfac=1.2"
GSR_DEEP_ORDER=1 GSR_DEEP_TAIL=0 GSR_DEEP_FACTOR_BWD=1.2 run "ordered tail=0 bwdfac=1.2"
GSR_DEEP_ORDER=0 GSR_DEEP_FACTOR_BWD=2.0 run "static bwdfac=2.0"
done 2>&1 | tee $out/ball.txt
