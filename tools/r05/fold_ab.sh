#!/bin/bash
# The compositing backward's sum over the wave's four rows (GSR_BWD_FOLD, raster_bwd.hip): permlane swaps (default)
# against the matrix pipe (fold1), LDS (fold2), LDS for the row's lanes as well (fold3).  Libraries from
# tools/r05/build_fold_variants.sh, loaded through GSR_LIBRARY.  Parity of every variant first, then the A/B.
out=${1:-gpurun_out/fold}; mkdir -p $out
for v in fold1 fold2 fold3; do
  echo "== parity $v"
  GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x \
    -k "rasterize_backward or tile16_matches or saturation or compositing or deep_tiles or depth_segment or determin or nan_cot or two_round" 2>&1 | tail -3
done 2>&1 | tee $out/parity.txt
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
  for v in default fold1 fold2 fold3; do
    if [ $v = default ]; then unset GSR_LIBRARY; else export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so; fi
    run "uniform $v"
    run "trained $v" --scene ply:$ply
    [ $rep = 1 ] && run "longtail $v" --scene longtail
  done
done 2>&1 | tee $out/steps.txt
