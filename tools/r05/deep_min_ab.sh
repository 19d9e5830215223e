#!/bin/bash
# GSR_DEEP_MIN 1024 (rounds 2-4) against 256 (round 5) on everything else the floor can touch: the long-tail bench
# scene, config 5's raster step (3 M / 4K, fused depth), config 3's training rate.   bash tools/r05/deep_min_ab.sh <out>
out=${1:-gpurun_out/deepmin}; mkdir -p $out
run() {
  local label=$1; shift
  python bench.py "$@" --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'x', k['raster_fwd']['calls_per_step'], 'bwd', k['raster_bwd']['ms'], 'x', k['raster_bwd']['calls_per_step'], 'tiles', d['config']['tile_list_length'])"
}
for min in 1024 256; do
  GSR_DEEP_MIN=$min run "longtail min=$min" --scene longtail
  GSR_DEEP_MIN=$min run "config5 min=$min" --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth
  GSR_DEEP_MIN=$min run "c2-200k min=$min" --gaussians 200000
  GSR_DEEP_MIN=$min run "dense-1M min=$min" --scale-lo 0.005 --scale-hi 0.05
done 2>&1 | tee $out/raster.txt
for min in 1024 256; do
  GSR_DEEP_MIN=$min python bench.py --train-only --train-iters 7000 --no-cogs 2>/dev/null | python -c "
import sys, json
t = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('train min=$min', 'config3', t['iters_per_s'], 'syncs', t.get('iters_per_s_with_caller_syncs'), 'fixed_1m', t['fixed_1m']['iters_per_s'], 'refined_1m', t['refined_1m']['iters_per_s'], 'full_res', t['full_resolution_from_step_0']['iters_per_s'], 'N_end', t['gaussians']['end'], 'by_res', {k: round(sum(v[x] for x in ('render','loss','backward','stats_exchange_optimizer')), 3) for k, v in t['phase_ms_median_by_resolution'].items()})"
done 2>&1 | tee $out/train.txt
