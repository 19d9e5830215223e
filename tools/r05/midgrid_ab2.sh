#!/bin/bash
# Grid size x split factors with the job order on: where between "every tile split" (small grids) and "only the
# longest" (1080p and up) the rule should sit, on the trained model rendered at several resolutions.
#   bash tools/r05/midgrid_ab2.sh gpurun_out/midgrid2
out=${1:-gpurun_out/midgrid2}; mkdir -p $out
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'mean', d['config']['tile_list_length']['mean'])"
}
export GSR_DEEP_ORDER_GRID=1100 GSR_SMALL_GRID=1100 GSR_DEEP_MIN=96
for res in "960 540" "1280 720" "1440 810" "1920 1080" "2560 1440"; do
  set -- $res
  S="--scene ply:$ply --width $1 --height $2"
  for fb in 0.25 0.35 0.5 0.7 1.0 1.4 2.0 3.0; do
    GSR_DEEP_FACTOR_BWD=$fb run "$1x$2 fwd 1.2 bwd $fb" $S
  done
  for ff in 0.15 0.3 0.6 0.9; do
    GSR_DEEP_FACTOR=$ff GSR_DEEP_FACTOR_BWD=1.0 run "$1x$2 fwd $ff bwd 1.0" $S
  done
done | tee $out/factors.txt
