#!/bin/bash
# Small grids (depth segments): no staged-ahead loads at all (noahead) / everywhere (committed) / not in the depth-segment
# instantiations, whose runs are a chunk or two long (aheadnoseg: -DGSR_STAGE_AHEAD_SEG=0).
out=$PWD/${1:-gpurun_out/aheadseg}; mkdir -p $out
ply=/tmp/config3_trained.ply
python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
train() {
  python bench.py --train-only --train-iters $2 --no-cogs 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t = d.get('train', d)
r = t.get('phase_ms_median_by_resolution') or {}
print('train($2) $1', 'it/s', t.get('iters_per_s'), {k: (v['render'], v['backward']) for k, v in r.items()})"
}
for rep in 1 2 3; do
  for v in noahead committed aheadnoseg; do
    export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so
    run "480x270 uniform200k $v" --gaussians 200000 --width 480 --height 270
    run "480x270 trained $v" --scene ply:$ply --width 480 --height 270
    train $v 2000
  done
done 2>&1 | tee $out/steps.txt
