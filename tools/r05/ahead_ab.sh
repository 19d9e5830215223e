#!/bin/bash
# The compositing loops with the next chunk's global loads in flight during the walk over the current one
# (GSR_STAGE_AHEAD, raster_common.h: default build) against stage_chunk in front of every chunk
# (libgsraster_noahead.so: both kernels built with -DGSR_STAGE_AHEAD=0).  Parity first, then the A/B.
out=${1:-gpurun_out/ahead}; mkdir -p $out
{ echo "== parity (default build = staged ahead)"
  timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x \
    -k "rasterize_forward or rasterize_backward or nd_rasterize or tile16_matches or saturation or compositing or deep_tiles or depth_segment or determin or nan_cot or two_round or job_order or alike or scan_mapping" 2>&1 | tail -3
  timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_render.py -q -x 2>&1 | tail -3
} 2>&1 | tee $out/parity.txt
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
  for v in noahead default; do
    if [ $v = default ]; then unset GSR_LIBRARY; else export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so; fi
    run "uniform $v"
    run "trained $v" --scene ply:$ply
    run "longtail $v" --scene longtail
    [ $rep = 1 ] && run "960x540 trained $v" --scene ply:$ply --width 960 --height 540
  done
done 2>&1 | tee $out/steps.txt
