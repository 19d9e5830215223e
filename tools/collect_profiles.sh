#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on a GPU box (run through gpurun):
#   tools/collect_profiles.sh <tag>        e.g. r01
# 1. kernel trace + stats of the default bench command  -> <tag>_kernel_stats.csv, <tag>_kernel_trace_summary.{json,txt}
# 2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy, LDS) -> <tag>_pmc_*.{json,txt}
# 3. traffic.json: HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction,
#    MI355X_MICROARCH.md HBM section)
# PMC passes never combine with other trace domains (only --kernel-trace).
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-pmc --train-iters 0 --event-every 0 --no-synced-regions"

rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $BENCH --steps 25 --warmup 5 > "$OUT/prof_kt.log" 2>&1
python "$ROOT/tools/summarize_prof.py" /tmp/prof_kt "$OUT/${TAG}_kernel_trace_summary.json" > "$OUT/${TAG}_kernel_trace_summary.txt"
cp "$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null
python "$ROOT/tools/step_seq.py" /tmp/prof_kt "$OUT/${TAG}_step_sequence.txt"
# the same step with the unchanged models' two host read-backs in the caller (vanilla_gs.py:784,811): where the GPU
# idles, with the tile lists built ahead of time on the side stream (default) and without (GSR_SPECULATE=0)
for SP in auto 0; do
  rm -rf /tmp/prof_sync
  GSR_SPECULATE=$SP timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sync -- $BENCH --steps 20 --warmup 5 --caller-syncs on --no-synced-regions > "$OUT/prof_sync_$SP.log" 2>&1
  python "$ROOT/tools/step_seq.py" /tmp/prof_sync "$OUT/${TAG}_step_sequence_caller_syncs_speculate_$SP.txt"
done

for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -- $BENCH --steps 4 --warmup 2 > "$OUT/prof_$C.log" 2>&1
  python "$ROOT/tools/summarize_prof.py" /tmp/prof_$C "$OUT/${TAG}_pmc_$C.json" > "$OUT/${TAG}_pmc_$C.txt"
done
rm -rf /tmp/prof_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq -- $BENCH --steps 4 --warmup 2 > "$OUT/prof_sq.log" 2>&1
python "$ROOT/tools/summarize_prof.py" /tmp/prof_sq "$OUT/${TAG}_pmc_sq.json" > "$OUT/${TAG}_pmc_sq.txt"
rm -rf /tmp/prof_lds
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/prof_lds -- $BENCH --steps 4 --warmup 2 > "$OUT/prof_lds.log" 2>&1
python "$ROOT/tools/summarize_prof.py" /tmp/prof_lds "$OUT/${TAG}_pmc_lds.json" > "$OUT/${TAG}_pmc_lds.txt"

python - "$OUT" "$TAG" <<'PY'
import json, sys
out, tag = sys.argv[1], sys.argv[2]
f = json.load(open(f"{out}/{tag}_pmc_FETCH_SIZE.json"))["counters"]
w = json.load(open(f"{out}/{tag}_pmc_WRITE_SIZE.json"))["counters"]
names = {"project_fwd": "project_fwd_kernel", "sh_fwd": "sh16_fwd_kernel", "count_reach": None,
         "raster_fwd": "raster_fwd_tile16_kernel", "raster_bwd": "raster_bwd_tile16_kernel",
         "sh_bwd": "sh16_bwd_kernel", "project_bwd": "project_bwd_kernel",
         "tile_scatter": "gsr_ts::scatter_kernel", "tile_rows": "tile_rows_kernel",
         "p2_rowcount": "gsr_p2::rowcount_kernel", "p2_emit": "gsr_p2::emit_kernel",
         "p2_colscatter": "gsr_p2::colscatter_kernel", "reach_records": "reach_records_kernel",
         "sort_scatter": "gsr_sort::scatter_kernel", "bsort_hist": "gsr_bsort::hist_kernel",
         "bsort_scan": "gsr_bsort::scan_kernel", "bsort_scatter": "gsr_bsort::scatter_kernel",
         "bsort_bucket_sort": "gsr_bsort::bucket_sort_kernel"}
t = {"_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
              f"({tag}_pmc_FETCH_SIZE.json, {tag}_pmc_WRITE_SIZE.json): KB units, FETCH_SIZE doubled on gfx950 as "
              "MI355X_MICROARCH.md#HBM prescribes (calibrated there for wide coalesced reads; the compositing "
              "kernels gather 4-32 B records, so their figure is an upper bound). Workload: bench.py default.",
     "_workload": {"gaussians": 1000000, "width": 1920, "height": 1080, "sh_degree": 3,
                   "scale_lo": 0.0025, "scale_hi": 0.025}}
for key, kn in names.items():
    if kn is None or kn not in f or kn not in w:
        continue
    fs, ws = f[kn]["FETCH_SIZE"]["mean"], w[kn]["WRITE_SIZE"]["mean"]
    t[key] = int((2 * fs + ws) * 1024)
    t[key + "_raw"] = {"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws}
json.dump(t, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in t.items() if not k.startswith("_") and not k.endswith("_raw")}))
PY
