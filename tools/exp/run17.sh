cd $GRAFT_REPO_ROOT
true
B="python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --scale-lo 0.005 --scale-hi 0.05 --gaussians 3000000 --width 3840 --height 2160 --render-depth --steps 20 --warmup 6"
show() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$1', d['ms_per_step'], d['ms_per_step_median'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e: print('$1 FAILED', e)"; }
{
GSR_TWO_ROUND=0 $B --fused-depth 2>gpurun_out/two_a.err | show c5_fused_single
GSR_TWO_ROUND=auto $B --fused-depth 2>gpurun_out/two_b.err | show c5_fused_two
GSR_TWO_ROUND=0 $B 2>/dev/null | show c5_single
GSR_TWO_ROUND=auto $B 2>gpurun_out/two_c.err | show c5_two
} > gpurun_out/r03_two_round_bench.txt 2>&1
tail -3 gpurun_out/two_b.err >> gpurun_out/r03_two_round_bench.txt
