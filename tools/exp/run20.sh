cd $GRAFT_REPO_ROOT
true
B="python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 40 --warmup 10"
C5="--scale-lo 0.005 --scale-hi 0.05 --gaussians 3000000 --width 3840 --height 2160 --render-depth"
show() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$1', d['ms_per_step'], d['ms_per_step_median'], d['config'].get('two_round_lists'), {k:(v['ms'], v['calls_per_step']) for k,v in d['kernels'].items() if k in ('bin_sorted','raster_fwd','raster_bwd','depth_order','count_reach')})
except Exception as e: print('$1 FAILED', e)"; }
{
GSR_TWO_ROUND=0 $B $C5 --fused-depth 2>/dev/null | show c5_fused_single
GSR_TWO_ROUND=auto $B $C5 --fused-depth 2>gpurun_out/two_b.err | show c5_fused_two
GSR_TWO_ROUND=0 $B $C5 2>/dev/null | show c5_single
GSR_TWO_ROUND=auto $B $C5 2>/dev/null | show c5_two
GSR_TWO_ROUND=auto $B 2>/dev/null | show default_auto
GSR_TWO_ROUND=auto $B --scale-lo 0.005 --scale-hi 0.05 2>/dev/null | show dense_auto
GSR_TWO_ROUND=0 $B --gaussians 20000000 --width 3840 --height 2160 --steps 8 --warmup 4 2>/dev/null | show 20M_single
GSR_TWO_ROUND=auto $B --gaussians 20000000 --width 3840 --height 2160 --steps 8 --warmup 4 2>/dev/null | show 20M_two
} > gpurun_out/r03_two_round_bench.txt 2>&1
tail -3 gpurun_out/two_b.err >> gpurun_out/r03_two_round_bench.txt
