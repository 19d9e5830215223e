cd $GRAFT_REPO_ROOT
L=gaussian-splatting-toolkit_amd/rasterizer/cuda
B="python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 100 --warmup 20 --event-every 5"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['ms_per_step'], d['ms_per_step_median'], 'bwd', d['kernels']['raster_bwd']['ms'], 'fwd', d['kernels']['raster_fwd']['ms'])"; }
{
$B 2>/dev/null | show base1
cp $L/libgsraster.so /tmp/base.so; cp $L/libgsraster_peff.so $L/libgsraster.so
$B 2>/dev/null | show peff1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -x -q -m gpu -k "rasterize_backward or golden or saturation or nan_cot or deep_tiles or deterministic or rgbd or inria" 2>&1 | tail -3
$B --scale-lo 0.005 --scale-hi 0.05 --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --steps 20 --warmup 5 2>/dev/null | show peff_c5
cp /tmp/base.so $L/libgsraster.so
$B 2>/dev/null | show base2
$B --scale-lo 0.005 --scale-hi 0.05 --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --steps 20 --warmup 5 2>/dev/null | show base_c5
} > gpurun_out/r03_peff.txt 2>&1
