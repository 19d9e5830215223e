cd $GRAFT_REPO_ROOT
for n in 10000000 20000000; do
timeout 600 python bench.py --gaussians $n --width 3840 --height 2160 --no-cpu-baseline --no-pmc --train-iters 0 --steps 5 --warmup 2 2>gpurun_out/big_$n.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$n', d['ms_per_step'], d['value'], d['config']['list_entries'], d['config']['intersections'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e: print('$n FAILED', e)"
rocm-smi --showmemuse 2>/dev/null | grep -i "vram\|GPU\[0\]" | head -3
done > gpurun_out/r03_big_scenes.txt 2>&1
tail -3 gpurun_out/big_*.err >> gpurun_out/r03_big_scenes.txt
