#!/bin/bash
# Two reached sub-tiles of a splat scheduled as ONE basic block (two independent dependency chains): forward (fwdpair),
# backward (bwdpair) against the committed build and the same sources without the pairing (pair00).
out=$PWD/${1:-gpurun_out/pair}; mkdir -p $out
for v in fwdpair bwdpair; do
  echo "== parity $v"
  GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x \
    -k "rasterize_forward or rasterize_backward or tile16_matches or saturation or compositing or deep_tiles or depth_segment or determin or nan_cot or two_round" 2>&1 | tail -3
done 2>&1 | tee $out/parity.txt
ply=/tmp/config3_trained.ply
python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
for rep in 1 2; do
  for v in committed pair00 fwdpair bwdpair; do
    export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so
    run "uniform $v"
    run "trained $v" --scene ply:$ply
    [ $rep = 1 ] && run "longtail $v" --scene longtail
    [ $rep = 1 ] && run "960x540 trained $v" --scene ply:$ply --width 960 --height 540
  done
done 2>&1 | tee $out/steps.txt
