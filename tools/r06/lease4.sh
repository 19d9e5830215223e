#!/bin/bash
# Lease 4: the whole -m gpu suite on the round's code; the overlap upper bound; blocking read-back round trips against the
# render mode per process; the regret table on the held-out families; one driver-form bench line.
out=$PWD/gpurun_out/lease4; mkdir -p $out
R=$PWD
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=20 2>&1 | tail -45 ) > $out/gpu_suite.txt
tail -3 $out/gpu_suite.txt
timeout 300 python tools/r06/overlap_probe.py > $out/overlap_probe.txt 2>&1; cat $out/overlap_probe.txt
run2() { # tag, env..., -- args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for i in $(seq 1 $REPS); do
    env "${envs[@]}" timeout 200 python tools/r06/mode480.py --tag $tag "$@" 2>>$out/err.log | grep '^{' >> $out/runs.jsonl
  done
}
REPS=8 run2 default X=1 -- --syncs 1
REPS=5 run2 nointerrupt HSA_ENABLE_INTERRUPT=0 -- --syncs 1
REPS=3 run2 nosync X=1 -- --syncs 0
python - <<PY | tee $out/round_trip.txt
import json
for l in open("$out/runs.jsonl"):
    r=json.loads(l); rt=r.get("round_trip_us_before_after")
    print(r["tag"], r["iters_per_s"], "render p50", r.get("render",{}).get("p50"), "bwd", r.get("backward",{}).get("p50"), "round trip us before", rt[0], "after", rt[1], "cpu", r.get("cpu_start_end"))
PY
timeout 1500 python tools/r06/regret.py $out/regret.txt > $out/regret_stdout.txt 2>&1; tail -20 $out/regret.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err; python - <<PY
import json
d=json.loads([l for l in open("$out/bench_driver_form.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,(dict,list))})
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list))})
PY
