#!/bin/bash
# Half-octave job order on / off, the validity fold on / off, backward split factor.
out=${1:-gpurun_out/lpt2}; mkdir -p $out
ply=/tmp/config3_trained.ply
if [ ! -f $ply ]; then
  python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
fi
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'lists', round(k['depth_order']['ms'] + k['bin_sorted']['ms'], 4))"
}
for rep in 1 2; do
  for ord in 1 0; do
    export GSR_DEEP_ORDER=$ord
    run "trained order=$ord" --scene ply:$ply
    run "uniform order=$ord"
    run "longtail order=$ord" --scene longtail
  done
done 2>&1 | tee $out/steps.txt
export GSR_DEEP_ORDER=1
for rep in 1 2; do
  GSR_LIBRARY=$PWD/tools/r05/libgsraster_nofold.so run "uniform nofold"
  run "uniform fold"
  GSR_LIBRARY=$PWD/tools/r05/libgsraster_nofold.so run "trained nofold" --scene ply:$ply
  run "trained fold" --scene ply:$ply
done 2>&1 | tee -a $out/steps.txt
for fb in 1.2 1.6 2.0; do
  GSR_DEEP_FACTOR_BWD=$fb run "trained bwdfac=$fb" --scene ply:$ply
  GSR_DEEP_FACTOR_BWD=$fb run "longtail bwdfac=$fb" --scene longtail
done 2>&1 | tee -a $out/steps.txt
for ord in 1 0; do
  export GSR_DEEP_ORDER=$ord
  run "config5 order=$ord" --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth
  run "dense-1M order=$ord" --scale-lo 0.005 --scale-hi 0.05
  run "c2-200k order=$ord" --gaussians 200000
done 2>&1 | tee -a $out/steps.txt
unset GSR_DEEP_ORDER
