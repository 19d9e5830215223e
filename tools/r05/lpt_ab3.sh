#!/bin/bash
# Why does the ordered launch cost the uniform scene 3 %?  Kernel trace both ways; tail splitting; split factors.
out=${1:-gpurun_out/lpt3}; mkdir -p $out
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
cd /tmp; export TMPDIR=/tmp
for ord in 1 0; do
  rm -rf /tmp/prof_$ord
  GSR_DEEP_ORDER=$ord rocprofv3 --kernel-trace --stats -d /tmp/prof_$ord --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > /dev/null 2>&1
  f=$(find /tmp/prof_$ord -name "*kernel_stats.csv" | head -1)
  echo "## kernel stats, uniform, GSR_DEEP_ORDER=$ord"; python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'].split('(')[0][-48:]:48s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done 2>&1 | tee $out/kernel_stats.txt
cd $GRAFT_REPO_ROOT
for tail in 0 4 8 12 16; do
  GSR_DEEP_TAIL=$tail run "uniform tail=$tail/64"
done 2>&1 | tee $out/steps.txt
for tail in 0 8; do
  GSR_DEEP_TAIL=$tail run "trained tail=$tail/64" --scene ply:$ply
  GSR_DEEP_TAIL=$tail run "dense-1M tail=$tail/64" --scale-lo 0.005 --scale-hi 0.05
done 2>&1 | tee -a $out/steps.txt
for fb in 2.0 3.0 4.0 6.0; do
  GSR_DEEP_FACTOR_BWD=$fb run "trained bwdfac=$fb" --scene ply:$ply
  GSR_DEEP_FACTOR_BWD=$fb run "longtail bwdfac=$fb" --scene longtail
done 2>&1 | tee -a $out/steps.txt
for f in 0.8 1.2 1.6 2.4; do
  GSR_DEEP_FACTOR=$f GSR_DEEP_FACTOR_BWD=2.0 run "trained fac=$f bwdfac=2.0" --scene ply:$ply
  GSR_DEEP_FACTOR=$f GSR_DEEP_FACTOR_BWD=2.0 run "longtail fac=$f bwdfac=2.0" --scene longtail
done 2>&1 | tee -a $out/steps.txt
