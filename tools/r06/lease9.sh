#!/bin/bash
# Lease 9: what changed after the closing suite -- held-out tests with the per-family bars, the bench contract tests (N > 1 affinity reset flag),
# and the trained model's parity figures in a driver-form line.
out=$PWD/gpurun_out/lease9; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_heldout.py tests/test_gpu_bench.py -q --durations=6 2>&1 | tail -25 > $out/tests.txt; tail -6 $out/tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cogs > $out/bench.json 2> $out/bench.err; python - <<PY
import json
d=json.loads([l for l in open("$out/bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"])
print({k:v for k,v in d["config"].items() if k.startswith(("trained","parity","render_480","config3","cpu_bind"))})
print(d["train"]["trained_raster"]["parity_vs_oracle"])
PY
