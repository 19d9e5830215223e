#!/bin/bash
# Deep tiles' extra sub-tile waves FIRST in block order (round 5) against behind the regular blocks (rounds 2-4,
# tools/r05/libgsraster_oldorder.so built with -DGSR_DEEP_EXTRAS_FIRST=0): wave timelines and step times.
out=${1:-gpurun_out/order}; mkdir -p $out
ply=/tmp/config3_trained.ply
if [ ! -f $ply ]; then
  python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
fi
old=$PWD/tools/r05/libgsraster_oldorder.so
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export GSR_LIBRARY=$old; else unset GSR_LIBRARY; fi
    run "trained $lib" --scene ply:$ply
    run "uniform $lib"
    run "longtail $lib" --scene longtail
    GSR_DEEP_MIN=1024 run "trained-min1024 $lib" --scene ply:$ply
  done
done 2>&1 | tee $out/steps.txt
unset GSR_LIBRARY
for min in 1024 256; do GSR_DEEP_MIN=$min run "c2-200k min=$min" --gaussians 200000; done 2>&1 | tee -a $out/steps.txt
for min in 256 1024; do GSR_DEEP_MIN=$min run "c2-200k min=$min" --gaussians 200000; done 2>&1 | tee -a $out/steps.txt
{
for lib in new old; do
  if [ $lib = old ]; then export GSR_LIBRARY=$old; else unset GSR_LIBRARY; fi
  echo "######## library: $lib"
  python tools/exp/wave_trace.py --scene ply:$ply 2>/dev/null
  python tools/exp/wave_trace.py --scene uniform 2>/dev/null
done
unset GSR_LIBRARY
echo "######## library: new, GSR_DEEP_MIN=1024 (round 4's floor)"
GSR_DEEP_MIN=1024 python tools/exp/wave_trace.py --scene ply:$ply 2>/dev/null
echo "######## library: new, long-tail scene"
python tools/exp/wave_trace.py --scene longtail 2>/dev/null
} > $out/wave_trace.txt 2>&1
cat $out/wave_trace.txt
