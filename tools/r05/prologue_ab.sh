#!/bin/bash
# The first chunk's loads in front of the per-pixel inputs (backward: speculatively at the end of the list; forward: in
# front of the resumed / segmented state) -- default build -- against the staged-ahead build without it (ahead1) and the
# build without any of it (noahead).  Parity first.
out=${1:-gpurun_out/prologue}; mkdir -p $out
{ echo "== parity (default build)"
  timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x \
    -k "rasterize_forward or rasterize_backward or nd_rasterize or tile16_matches or saturation or compositing or deep_tiles or depth_segment or determin or nan_cot or two_round or job_order or alike or scan_mapping" 2>&1 | tail -3
  timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_render.py tests/test_gpu_cogs.py -q -x 2>&1 | tail -3
} 2>&1 | tee $out/parity.txt
ply=/tmp/config3_trained.ply
python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train_default.json 2> $out/train.err || exit 1
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
train() {
  python bench.py --train-only --train-iters 7000 --no-cogs 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t = d.get('train', d)
r = t.get('phase_ms_median_by_resolution') or {}
print('train $1', 'it/s', t.get('iters_per_s'), 'syncs', t.get('iters_per_s_with_caller_syncs'), {k: (v['render'], v['backward']) for k, v in r.items()}, 'fixed_1m', (t.get('fixed_1m') or {}).get('iters_per_s'), 'refined_1m', (t.get('refined_1m') or {}).get('iters_per_s'))"
}
for rep in 1 2; do
  for v in ahead1 default; do
    if [ $v = default ]; then unset GSR_LIBRARY; else export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so; fi
    run "uniform $v"
    run "trained $v" --scene ply:$ply
    run "960x540 trained $v" --scene ply:$ply --width 960 --height 540
    run "480x270 trained $v" --scene ply:$ply --width 480 --height 270
    [ $rep = 1 ] && run "longtail $v" --scene longtail
    train $v
  done
done 2>&1 | tee $out/steps.txt
