#!/bin/bash
# The middle stage of the reference's coarse-to-fine schedule (960 x 540 = 2 040 tiles; 480 x 270 = 510): the trained
# model's compositing at those sizes and config 3's rate under variants of the small-grid rules and the job order.
#   bash tools/r05/midgrid_ab.sh gpurun_out/midgrid
out=${1:-gpurun_out/midgrid}; mkdir -p $out
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
tail -1 $out/train_default.json
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'lists', round(k['depth_order']['ms'] + k['bin_sorted']['ms'], 4), 'tiles', d['config']['tile_list_length'])"
}
for res in "960 540" "480 270"; do
  set -- $res
  S="--scene ply:$ply --width $1 --height $2"
  run "$1x$2 default" $S
  GSR_SMALL_GRID_BWD=2560 run "$1x$2 bwd-splits-every-tile" $S
  GSR_DEEP_ORDER_GRID=1100 run "$1x$2 ordered" $S
  GSR_DEEP_ORDER_GRID=1100 GSR_DEEP_FACTOR_BWD=1.0 run "$1x$2 ordered bwd-factor-1.0" $S
  GSR_DEEP_ORDER_GRID=1100 GSR_DEEP_FACTOR_BWD=0.5 run "$1x$2 ordered bwd-factor-0.5" $S
  GSR_DEEP_ORDER_GRID=1100 GSR_SMALL_GRID_BWD=2560 run "$1x$2 ordered bwd-splits-every-tile" $S
  GSR_DEEP_ORDER_GRID=1100 GSR_SMALL_GRID=1100 run "$1x$2 ordered as-a-large-grid" $S
  GSR_DEPTH_SEGMENTS_GRID=2560 GSR_SMALL_GRID_BWD=2560 run "$1x$2 depth-segments" $S
  GSR_DEPTH_SEGMENTS_GRID=2560 GSR_SMALL_GRID_BWD=2560 GSR_DEPTH_SEGMENTS=4 GSR_DEPTH_SEGMENTS_FWD=4 run "$1x$2 depth-segments-4" $S
done | tee $out/trained_midgrid.txt
for v in "GSR_DEEP_ORDER_GRID=1100" "GSR_DEEP_ORDER_GRID=1100 GSR_DEEP_FACTOR_BWD=1.0" "GSR_SMALL_GRID_BWD=2560" "GSR_DEPTH_SEGMENTS_GRID=2560 GSR_SMALL_GRID_BWD=2560 GSR_DEPTH_SEGMENTS=4 GSR_DEPTH_SEGMENTS_FWD=4" "GSR_NOTHING=1"; do
  echo "config3 $v: $(env $v python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1)"
done | tee $out/config3_variants.txt
