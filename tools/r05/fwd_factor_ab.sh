#!/bin/bash
# Forward split factor on grids above 1080p (the backward's follows the grid: does the forward's want to?).
out=${1:-gpurun_out/fwdfactor}; mkdir -p $out
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'mean', d['config']['tile_list_length']['mean'])"
}
for res in "2560 1440" "3840 2160"; do
  set -- $res
  for scene in uniform ball longtail "ply:$ply"; do
    for ff in 1.2 2.0 3.0 5.0; do
      for tail in 8 0; do
        GSR_DEEP_FACTOR=$ff GSR_DEEP_TAIL=$tail run "$scene $1x$2 fwd-factor $ff tail $tail" --scene $scene --width $1 --height $2
      done
    done
  done
done | tee $out/fwd_factor.txt
for ff in 1.2 3.0; do
  GSR_DEEP_FACTOR=$ff run "config5-3M 4K fwd-factor $ff" --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth
done | tee -a $out/fwd_factor.txt
