#!/bin/bash
# The staged-ahead prologue guarded for empty lists (default build) against the build of the closing records (committed):
# parity subset, then the A/B (the guard is one wave-uniform branch per wave).
out=$PWD/${1:-gpurun_out/guard}; mkdir -p $out
{ timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "rasterize_forward or rasterize_backward or nd_rasterize or tile16_matches or saturation or compositing or deep_tiles or depth_segment or determin or nan_cot or two_round or job_order or alike" 2>&1 | tail -2
  timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_render.py tests/test_gpu_cogs.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -2
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} 2>&1 | tee $out/parity.txt
ply=/tmp/config3_trained.ply
python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply 2> $out/train.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t = d.get('train', d)
print('train default it/s', t.get('iters_per_s'), 'syncs', t.get('iters_per_s_with_caller_syncs'))" | tee $out/steps.txt
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
for rep in 1 2; do
  for v in committed default; do
    if [ $v = default ]; then unset GSR_LIBRARY; else export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so; fi
    run "uniform $v"
    run "trained $v" --scene ply:$ply
    [ $rep = 1 ] && run "960x540 trained $v" --scene ply:$ply --width 960 --height 540
  done
done 2>&1 | tee -a $out/steps.txt
