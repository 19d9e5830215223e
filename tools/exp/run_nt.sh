#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gaussian-splatting-toolkit_amd/rasterizer/cuda/libgsraster.so
for rep in 1 2 3; do for v in plain nt; do
  cp tools/exp/tmp/lib_$v.so $L
  timeout 300 python bench.py --no-pmc --no-cpu-baseline --train-iters 0 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={n: round(v if isinstance(v,(int,float)) else v.get('ms',0),4) for n,v in d['kernels'].items()}
print('$v', d['ms_per_step'], {a:k[a] for a in ('raster_fwd','raster_bwd')})"
done; done
cp tools/exp/tmp/lib_nt.so $L
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -x -q -k "raster or nan or golden or deep" 2>&1 | tail -2
