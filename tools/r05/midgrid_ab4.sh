#!/bin/bash
# Fixed backward split factors (x mean list length, job order on) on grids below 4 096 tiles, per distribution: what
# the floor of the grid-scaled factor should be.
#   bash tools/r05/midgrid_ab4.sh gpurun_out/midgrid4
out=${1:-gpurun_out/midgrid4}; mkdir -p $out
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'tiles', d['config']['tile_list_length'])"
}
export GSR_DEEP_FACTOR_BWD_SCALED=0 GSR_DEEP_ORDER_GRID=1100
for res in "960 540" "1280 720"; do
  set -- $res
  for scene in "uniform" "ball" "longtail" "ply:$ply"; do
    for fb in 0.5 0.7 1.0 1.2 1.4 1.7 2.0; do
      GSR_DEEP_FACTOR_BWD=$fb run "$scene $1x$2 bwd $fb" --scene $scene --width $1 --height $2
    done
  done
  for fb in 0.5 1.0 1.4 2.0; do
    GSR_DEEP_FACTOR_BWD=$fb run "uniform-200k $1x$2 bwd $fb" --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05 --width $1 --height $2
  done
done | tee $out/floor.txt
