#!/bin/bash
# The fast stable job order with its round-5 defaults against the static map, every workload, three repetitions.
out=${1:-gpurun_out/lpt5}; mkdir -p $out
ply=/tmp/config3_trained.ply
if [ ! -f $ply ]; then
  python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
fi
run() {
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'lists', round(k['depth_order']['ms'] + k['bin_sorted']['ms'], 4))"
}
for rep in 1 2 3; do
  for ord in 1 0; do
    export GSR_DEEP_ORDER=$ord
    run "uniform order=$ord"
    run "trained order=$ord" --scene ply:$ply
    run "c2-200k order=$ord" --gaussians 200000
  done
done 2>&1 | tee $out/steps.txt
for ord in 1 0; do
  export GSR_DEEP_ORDER=$ord
  run "longtail order=$ord" --scene longtail
  run "config5 order=$ord" --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth
  run "dense-1M order=$ord" --scale-lo 0.005 --scale-hi 0.05
  run "small-grid order=$ord" --gaussians 300000 --width 480 --height 270 --scale-lo 0.005 --scale-hi 0.03
done 2>&1 | tee -a $out/steps.txt
export GSR_DEEP_ORDER=1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_s; rocprofv3 --kernel-trace --stats -d /tmp/prof_s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > /dev/null 2>&1
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/$out/kernel_stats.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'].split('(')[0][-48:]:48s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
cd $GRAFT_REPO_ROOT
for ord in 1 0; do
  GSR_DEEP_ORDER=$ord python bench.py --train-only --train-iters 7000 --no-cogs 2>/dev/null | python -c "
import sys, json
t = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('train order=$ord', 'config3', t['iters_per_s'], 'syncs', t.get('iters_per_s_with_caller_syncs'), 'fixed_1m', t['fixed_1m']['iters_per_s'], 'refined_1m', t['refined_1m']['iters_per_s'], 'full_res', t['full_resolution_from_step_0']['iters_per_s'], 'one_op', t['one_op_path']['iters_per_s'], 'by_res', {k: round(sum(v[x] for x in ('render','loss','backward','stats_exchange_optimizer')), 3) for k, v in t['phase_ms_median_by_resolution'].items()})"
done 2>&1 | tee $out/train.txt
