#!/bin/bash
# Forward compositing loop: the three LDS records of a splat read where they are used (default: two dependent round
# trips per splat), together (fwdlds1), one splat ahead (fwdlds2) -- raster_fwd.hip: GSR_FWD_LDS_AHEAD.
# noahead = the build without the staged-ahead global loads (GSR_STAGE_AHEAD=0), for the record of that change.
out=${1:-gpurun_out/fwdlds}; mkdir -p $out
for v in fwdlds1 fwdlds2; do
  echo "== parity $v"
  GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x \
    -k "rasterize_forward or nd_rasterize or tile16_matches or compositing or deep_tiles or depth_segment or two_round or job_order or scan_mapping" 2>&1 | tail -3
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
  for v in noahead default fwdlds1 fwdlds2; do
    if [ $v = default ]; then unset GSR_LIBRARY; else export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so; fi
    run "uniform $v"
    run "trained $v" --scene ply:$ply
    run "960x540 trained $v" --scene ply:$ply --width 960 --height 540
    [ $rep = 1 ] && run "longtail $v" --scene longtail
  done
done 2>&1 | tee $out/steps.txt
