#!/bin/bash
# backward kernel built for 5 / 6 waves per SIMD (96 / 80 VGPRs, 9 / 26 spilled) against the default (110 VGPRs, 4 waves)
out=${1:-gpurun_out/occ}; mkdir -p $out
ply=/tmp/config3_trained.ply
if [ ! -f $ply ]; then
  python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
fi
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
for rep in 1 2; do
  for v in default bwdw5 bwdw6; do
    if [ $v = default ]; then unset GSR_LIBRARY; else export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so; fi
    run "uniform $v"
    run "trained $v" --scene ply:$ply
    run "longtail $v" --scene longtail
  done
done 2>&1 | tee $out/steps.txt
