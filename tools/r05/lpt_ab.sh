#!/bin/bash
# Longest-job-first block order (GSR_DEEP_ORDERED, round 5) on / off: step and compositing times, wave timelines.
out=${1:-gpurun_out/lpt}; mkdir -p $out
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
    run "c2-200k order=$ord" --gaussians 200000
  done
done 2>&1 | tee $out/steps.txt
for ord in 1 0; do
  export GSR_DEEP_ORDER=$ord
  run "config5 order=$ord" --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth
  run "dense-1M order=$ord" --scale-lo 0.005 --scale-hi 0.05
  for fac in 0.8 1.2 1.6 2.4; do GSR_DEEP_FACTOR=$fac run "trained order=$ord fac=$fac" --scene ply:$ply; done
  for fac in 0.8 1.6; do GSR_DEEP_FACTOR=$fac run "longtail order=$ord fac=$fac" --scene longtail; done
done 2>&1 | tee -a $out/steps.txt
{
for ord in 1 0; do
  export GSR_DEEP_ORDER=$ord
  echo "######## GSR_DEEP_ORDER=$ord"
  python tools/exp/wave_trace.py --scene ply:$ply 2>/dev/null
  python tools/exp/wave_trace.py --scene uniform 2>/dev/null
  python tools/exp/wave_trace.py --scene longtail 2>/dev/null
done
} > $out/wave_trace.txt 2>&1
unset GSR_DEEP_ORDER
grep "==\|span\|GSR_DEEP_ORDER\|scene" $out/wave_trace.txt
