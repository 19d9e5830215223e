#!/bin/bash
# co-gs training leg (3 M Gaussians, 4K, RGB + depth compositing) with and without the staged-ahead loads
out=$PWD/${1:-gpurun_out/cogs_ab}; mkdir -p $out
for rep in 1 2; do
  for v in noahead committed; do
    export GSR_LIBRARY=$PWD/tools/r05/libgsraster_$v.so
    python bench.py --train-only --train-iters 600 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t = d.get('train', d); c = t.get('cogs_3m_4k') or {}
print('cogs $v', 'it/s', c.get('iters_per_s'), c.get('phase_ms_median'), 'config3(600)', t.get('iters_per_s'))"
  done
done 2>&1 | tee $out/cogs_ab.txt
