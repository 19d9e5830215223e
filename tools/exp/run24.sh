cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 40 --warmup 10"
show() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$1', d['ms_per_step'], d['ms_per_step_median'], d['config'].get('two_round_lists'), d['config']['list_entries'])
except Exception as e: print('$1 FAILED', e)"; }
{
GSR_TWO_ROUND=0 $B --scale-lo 0.005 --scale-hi 0.05 2>/dev/null | show dense_single
GSR_TWO_ROUND=1 $B --scale-lo 0.005 --scale-hi 0.05 2>/dev/null | show dense_two
GSR_TWO_ROUND=0 $B 2>/dev/null | show default_single
GSR_TWO_ROUND=1 $B 2>/dev/null | show default_two
GSR_TWO_ROUND=0 $B --gaussians 3000000 --scale-lo 0.005 --scale-hi 0.05 2>/dev/null | show 3M_1080p_single
GSR_TWO_ROUND=auto $B --gaussians 3000000 --scale-lo 0.005 --scale-hi 0.05 2>/dev/null | show 3M_1080p_auto
} > gpurun_out/r03_two_round_threshold.txt 2>&1
