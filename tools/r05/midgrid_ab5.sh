#!/bin/bash
# After the device-side guard for lists that are all alike: old rule (fixed factor, order from 2 560 tiles) vs new.
out=${1:-gpurun_out/midgrid5}; mkdir -p $out
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
for res in "960 540" "1280 720" "1920 1080"; do
  set -- $res
  ab "uniform-1M $1x$2" --width $1 --height $2
  ab "uniform-200k $1x$2" --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05 --width $1 --height $2
  ab "trained $1x$2" --scene ply:$ply --width $1 --height $2
done
ab "ball-1M 1920x1080" --scene ball
ab "longtail-1M 1920x1080" --scene longtail
ab "uniform-1M 2560x1440" --width 2560 --height 1440
} | tee $out/alike_ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "job_order or raster or deep or segment" 2>&1 | tail -3
