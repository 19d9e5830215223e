cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in fused-render graph; do
  rm -rf /tmp/kt_$mode
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$mode -- python $R/tools/train_bench.py --gaussians 1000000 --iters 120 --$mode > /tmp/kt_$mode.log 2>&1
  echo "$mode $(tail -1 /tmp/kt_$mode.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["iters_per_s"],1), "it/s under the tracer")' 2>/dev/null)"
  python $R/tools/exp/gap_stats.py /tmp/kt_$mode
done > $R/gpurun_out/r03_graph_gaps.txt 2>&1
