#!/bin/bash
# config 3's training rate (every reference default: 480x270 -> 960x540 -> 1080p) with and without the staged-ahead loads
out=${1:-gpurun_out/ahead_train}; mkdir -p $out
for rep in 1 2; do
  for v in noahead default; do
    if [ $v = default ]; then unset GSR_LIBRARY; else export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so; fi
    python bench.py --train-only --train-iters 7000 --no-cogs 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t = d.get('train', d)
r = t.get('phase_ms_median_by_resolution') or {}
print('$v', 'it/s', t.get('iters_per_s'), 'syncs', t.get('iters_per_s_with_caller_syncs'), {k: (v['render'], v['backward']) for k, v in r.items()}, 'fixed_1m', (t.get('fixed_1m') or {}).get('iters_per_s'), 'refined_1m', (t.get('refined_1m') or {}).get('iters_per_s'))"
  done
done 2>&1 | tee $out/train_ab.txt
