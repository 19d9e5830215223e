#!/bin/bash
# A/B of the deep-tile rule on the TRAINED distribution (the model config 3 ends with) and on the uniform default:
# GSR_DEEP_MIN x GSR_DEEP_FACTOR -> step / compositing forward / backward ms.  Run on the GPU box:
#   bash tools/r05/trained_ab.sh gpurun_out/ab
out=${1:-gpurun_out/ab}; mkdir -p $out
ply=/tmp/config3_trained.ply
if [ ! -f $ply ]; then
  python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
fi
run() {  # label, scene args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'lists', round(k['depth_order']['ms'] + k['bin_sorted']['ms'], 4), 'tiles', d['config']['tile_list_length'])"
}
for min in 1024 512 384 256 192 128 64; do
  for fac in 1.2 1.6 2.0 3.0; do
    GSR_DEEP_MIN=$min GSR_DEEP_FACTOR=$fac run "trained min=$min fac=$fac" --scene ply:$ply
  done
done | tee $out/trained.txt
for min in 1024 256 128; do
  for fac in 1.2 2.0 3.0; do
    GSR_DEEP_MIN=$min GSR_DEEP_FACTOR=$fac run "uniform min=$min fac=$fac"
  done
done | tee $out/uniform.txt
