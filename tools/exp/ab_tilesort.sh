# A/B of the list builders through bench.py: tools/exp/ab_tilesort.sh  (GSR_TILE_SORT=s single pass, t two-level, - default)
for args in "" "--gaussians 10000 --width 256 --height 256 --sh-degree 0 --scale-lo 0.005 --scale-hi 0.05" "--gaussians 100000" "--gaussians 200000 --scale-lo 0.005 --scale-hi 0.05" "--gaussians 500000"; do
for m in s t -; do
if [ $m = - ]; then unset GSR_TILE_SORT; else export GSR_TILE_SORT=$m; fi
python bench.py --no-pmc --no-cpu-baseline $args | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', '$args', d['ms_per_step'], d['ms_per_step_median'], d['config'].get('list_entries'), {k:v['ms'] for k,v in d['kernels'].items() if k in ('count_reach','depth_order','bin_sorted')})"
done; done
