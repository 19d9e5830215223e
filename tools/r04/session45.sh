#!/bin/bash
cd $GRAFT_REPO_ROOT
D=gaussian-splatting-toolkit_amd/rasterizer/cuda
echo "== with atomics"; BLOB=1 SEGS=16 python tools/exp/seg_ab.py 480 270 300000 30 2>&1 | grep "segments"
cp $D/libgsraster.so /tmp/keep.so; cp $D/libgsraster_noatom.so $D/libgsraster.so
echo "== gradient atomics compiled out (results meaningless)"; BLOB=1 SEGS=16 python tools/exp/seg_ab.py 480 270 300000 30 2>&1 | grep "segments" | cut -c1-120
echo "== 1080p default scene, atomics compiled out"; python bench.py --no-pmc --no-cpu-baseline --train-iters 0 --no-synced-regions --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {n:round(v['ms'],4) for n,v in d['kernels'].items() if 'raster' in n})"
cp /tmp/keep.so $D/libgsraster.so
