#!/bin/bash
# Forward depth segments 8 / 12 / 16 on 510-tile grids, per distribution, then config 3's rate twice each.
out=gpurun_out/smallgrid; mkdir -p $out
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'mean', d['config']['tile_list_length']['mean'])"
}
for scene in "uniform" "ball" "longtail" "ply:$ply"; do
  for n in 300000 1000000; do
    for f in 4 8 12 16; do
      GSR_DEPTH_SEGMENTS_FWD=$f run "$scene n=$n 480x270 fwd-segments $f" --scene $scene --gaussians $n --width 480 --height 270
    done
  done
done | tee $out/fwd_segments.txt
for v in "GSR_DEPTH_SEGMENTS_FWD=16" "GSR_DEPTH_SEGMENTS_FWD=16 GSR_DEPTH_SEGMENTS_MIN=256" "GSR_DEPTH_SEGMENTS_FWD=16 GSR_DEPTH_SEGMENTS_MIN=384" "GSR_NOTHING=1" "GSR_DEPTH_SEGMENTS_FWD=16"; do
  echo "config3 $v: $(env $v python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-330)"
done | tee $out/config3_segments2.txt
