cd $GRAFT_REPO_ROOT
for w in 4 2 1; do
  GSR_SH_WAVES=$w python bench.py --no-cpu-baseline --no-pmc --train-iters 0 --steps 60 --warmup 10 --event-every 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('waves $w', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if k.startswith('sh')})"
done > gpurun_out/r03_sh_waves.txt 2>&1
