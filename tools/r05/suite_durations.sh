#!/bin/bash
# The GPU suite with its slowest tests listed, and a repeated A/B of the smallest uniform case.
out=gpurun_out; mkdir -p $out
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
for i in 1 2 3; do
  GSR_DEEP_FACTOR_BWD_SCALED=0 GSR_DEEP_ORDER_GRID=2560 run "uniform-200k 960x540 old" --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05 --width 960 --height 540
  run "uniform-200k 960x540 new" --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05 --width 960 --height 540
done | tee $out/uniform200k_repeat.txt
timeout 2000 python -m pytest tests -q -m gpu --durations=45 > $out/gpu_suite_durations.log 2>&1
tail -60 $out/gpu_suite_durations.log
