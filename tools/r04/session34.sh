#!/bin/bash
cd $GRAFT_REPO_ROOT
Q="--no-pmc --no-cpu-baseline --train-iters 0 --no-synced-regions --steps 60"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={n:round(v['ms'],4) for n,v in d['kernels'].items() if 'raster' in n}
print('$1', d['ms_per_step'], d.get('ms_per_step_median'), k)"; }
for s in 1 4 8; do
  GSR_DEPTH_SEGMENTS=$s GSR_DEPTH_SEGMENTS_GRID=100000 python bench.py $Q --scene longtail 2>/dev/null | show "longtail segs=$s"
done
for s in 1 4; do
  GSR_DEPTH_SEGMENTS=$s GSR_DEPTH_SEGMENTS_GRID=100000 python bench.py $Q 2>/dev/null | show "default segs=$s"
done
