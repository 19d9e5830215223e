#!/bin/bash
# Forward-only depth segments for the longest lists on grids above the small ones (GSR_DEPTH_SEGMENTS_FWD_GRID).
out=gpurun_out/fwdseg; mkdir -p $out
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'parity', (d.get('parity_vs_oracle') or {}).get('img_max_abs_stable'))"
}
{
for sc in "trained-960x540 --scene ply:$ply --width 960 --height 540" "trained-1080p --scene ply:$ply" "trained-1280x720 --scene ply:$ply --width 1280 --height 720" "longtail-1080p --scene longtail" "ball-1080p --scene ball" "uniform-1080p --scene uniform"; do
  set -- $sc; label=$1; shift
  run "$label off" "$@"
  for runs in 2 4 8; do
    for fac in 1.5 3.0 5.0; do
      GSR_DEPTH_SEGMENTS_FWD_GRID=9000 GSR_DEPTH_SEGMENTS_FWD_RUNS=$runs GSR_DEPTH_SEGMENTS_FWD_FACTOR=$fac run "$label runs $runs factor $fac" "$@"
    done
  done
done
} | tee $out/fwdseg_ab.txt
