#!/bin/bash
# split jobs keyed by length / 16: every workload, ordered against static
out=${1:-gpurun_out/lpt6}; mkdir -p $out
ply=/tmp/config3_trained.ply
if [ ! -f $ply ]; then
  python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
fi
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
for rep in 1 2; do
  for ord in 1 0; do
    export GSR_DEEP_ORDER=$ord
    run "uniform order=$ord"
    run "trained order=$ord" --scene ply:$ply
    run "ball order=$ord" --scene ball
    run "longtail order=$ord" --scene longtail
    run "dense-1M order=$ord" --scale-lo 0.005 --scale-hi 0.05
  done
done 2>&1 | tee $out/steps.txt
for ord in 1 0; do
  GSR_DEEP_ORDER=$ord python tools/train_bench.py --iters 400 --sh-interval 100 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fixed_1m-like order=$ord it/s', round(d['iters_per_s'], 1))"
done 2>&1 | tee -a $out/steps.txt
