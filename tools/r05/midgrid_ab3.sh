#!/bin/bash
# The grid-scaled backward split factor (GSR_DEEP_FACTOR_BWD_SCALED, job order from 1 100 tiles) against the fixed one
# (and the old ordering limit) on the other distributions and sizes, and on the training legs.
#   bash tools/r05/midgrid_ab3.sh gpurun_out/midgrid3
out=${1:-gpurun_out/midgrid3}; mkdir -p $out
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
tail -1 $out/train_default.json
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'mean', d['config']['tile_list_length']['mean'])"
}
ab() {  # label, args...
  local label=$1; shift
  GSR_DEEP_FACTOR_BWD_SCALED=0 GSR_DEEP_ORDER_GRID=2560 run "$label  old" "$@"
  run "$label  new" "$@"
}
{
for res in "960 540" "1280 720" "2560 1440" "3840 2160"; do
  set -- $res
  ab "uniform-1M $1x$2" --width $1 --height $2
  ab "ball-1M $1x$2" --scene ball --width $1 --height $2
  ab "longtail-1M $1x$2" --scene longtail --width $1 --height $2
done
ab "trained 3840x2160" --scene ply:$ply --width 3840 --height 2160
ab "config5-3M 3840x2160" --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth
ab "uniform-200k 1280x720" --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05 --width 1280 --height 720
} | tee $out/scaled_ab.txt
for v in "GSR_DEEP_FACTOR_BWD_SCALED=0 GSR_DEEP_ORDER_GRID=2560" "GSR_NOTHING=1"; do
  echo "train-legs $v: $(env $v python bench.py --train-only --train-iters 7000 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: (d[k].get('iters_per_s') if isinstance(d[k], dict) else d[k]) for k in ('iters_per_s', 'iters_per_s_with_caller_syncs', 'full_resolution_from_step_0', 'refined_1m', 'fixed_1m', 'cogs_3m_4k', 'one_op_path') if k in d}, d['phase_ms_median_by_resolution'])")"
done | tee $out/train_legs.txt
