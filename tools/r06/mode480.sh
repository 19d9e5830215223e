#!/bin/bash
# VERDICT r5 item 1, "cause first": the 480x270 render phase in fresh processes, one variant at a time, then a
# kernel + HIP-API trace of one process for the timeline (stream ids, inter-kernel gaps, which host call blocks).
out=$PWD/gpurun_out/mode480; mkdir -p $out
R=$PWD
run() { # tag, env..., -- args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for i in $(seq 1 $REPS); do
    env "${envs[@]}" timeout 200 python tools/r06/mode480.py --tag $tag "$@" 2>>$out/err.log | grep '^{' >> $out/runs.jsonl
  done
}
REPS=${REPS:-6}
run default_syncs X=1 -- --syncs 1
run default_nosync X=1 -- --syncs 0
run nospec_syncs GSR_SPECULATE=0 -- --syncs 1
run nospec_nosync GSR_SPECULATE=0 -- --syncs 0
REPS=3
run pollspin_syncs GSR_POLL_YIELD=0 -- --syncs 1
run calib_syncs X=1 -- --syncs 1 --calib 1
run fused_nosync X=1 -- --syncs 0 --fused 1
run graph_nosync X=1 -- --syncs 0 --graph 1
run noseg_syncs GSR_DEPTH_SEGMENTS=1 -- --syncs 1
python - <<PY > $out/summary.txt
import json, collections
rows=[json.loads(l) for l in open("$out/runs.jsonl")]
by=collections.OrderedDict()
for r in rows: by.setdefault(r["tag"],[]).append(r)
print("tag                 n  it/s(each)                                render p50 (each)                      slow share (each)")
for t,rs in by.items():
    print("%-18s %2d  %-42s %-38s %s" % (t,len(rs)," ".join("%.0f"%r["iters_per_s"] for r in rs)," ".join("%.3f"%r["render"]["p50"] for r in rs if "render" in r)," ".join("%.2f"%r.get("render_slow_share",-1) for r in rs)))
for r in rows[:4]+rows[-2:]:
    print(r["tag"], r.get("render_block_medians"), r.get("render_hist_ms"))
PY
cat $out/summary.txt
# timeline of one process (kernel + HIP API trace); keep the raw CSV window small
cd /tmp && export TMPDIR=/tmp
for v in syncs1 syncs0; do
  rm -rf /tmp/prof_m
  s=1; [ $v = syncs0 ] && s=0
  timeout 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/prof_m -- python $R/tools/r06/mode480.py --iters 400 --syncs $s --tag traced_$v > $out/traced_$v.json 2>>$out/err.log
  python $R/tools/r06/timeline.py /tmp/prof_m $out/timeline_$v.txt $out/timeline_$v.csv.gz >> $out/err.log 2>&1
done
ls -la $out
