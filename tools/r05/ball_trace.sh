#!/bin/bash
out=${1:-gpurun_out/ball}; mkdir -p $out
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'tiles', d['config']['tile_list_length'])"
}
{
for ord in 0 1; do
  GSR_DEEP_ORDER=$ord run "ball order=$ord" --scene ball
  echo "######## GSR_DEEP_ORDER=$ord"
  GSR_DEEP_ORDER=$ord python tools/exp/wave_trace.py --scene ball 2>/dev/null
done
GSR_DEEP_ORDER=1 GSR_DEEP_TAIL=0 GSR_DEEP_FACTOR_BWD=1.2 run "ball order=1 tail=0 bwdfac=1.2" --scene ball
GSR_DEEP_ORDER=0 GSR_DEEP_FACTOR_BWD=1.2 run "ball order=0 bwdfac=1.2" --scene ball
} 2>&1 | tee $out/ball_trace.txt
