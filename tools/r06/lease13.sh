#!/bin/bash
# Lease 13: XCD map in 4 x 2 blocks on grids of up to 1 100 tiles -- parity, then the A/B against the previous library.
out=$PWD/gpurun_out/lease13; mkdir -p $out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_render.py tests/test_gpu_cogs.py tests/test_gpu_heldout.py -q 2>&1 | tail -4 | tee $out/parity.txt
young=/tmp/config3_young.ply
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2> $out/train.err
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'sum', round(k['raster_fwd']['ms'] + k['raster_bwd']['ms'], 4))"
}
prev=$R/gaussian-splatting-toolkit_amd/rasterizer/cuda/libgsraster_prev.so
for args in "ply:$young 480 270" "uniform 480 270" "room 480 270" "ball 480 270" "uniform 640 360" "room 400 300"; do
  set -- $args
  GSR_LIBRARY=$prev run "8x4 $1 $2x$3" --scene $1 --gaussians 300000 --width $2 --height $3
  run "4x2 $1 $2x$3" --scene $1 --gaussians 300000 --width $2 --height $3
done 2>&1 | tee $out/xcd_blocks_ab.txt
for i in 1 2 3; do
  python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-330 | sed 's/^/4x2: /' | tee -a $out/config3.txt
  GSR_LIBRARY=$prev python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-330 | sed 's/^/8x4: /' | tee -a $out/config3.txt
done
