#!/bin/bash
# Round 3, verdict item 4: where do the compositing backward's idle issue slots go?
#   tools/collect_attribution.sh <tag>   (through gpurun) -> gpurun_out/<tag>_pmc_wait*.{json,txt}, <tag>_counters_avail.txt
# Separate --pmc passes (kernel trace only), a few counters each.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-pmc --train-iters 0 --event-every 0 --steps 4 --warmup 2"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > "$OUT/${TAG}_counters_avail.txt"
pass() { name=$1; shift; rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/prof_$name -- $BENCH > "$OUT/prof_$name.log" 2>&1
  python "$ROOT/tools/summarize_prof.py" /tmp/prof_$name "$OUT/${TAG}_pmc_$name.json" > "$OUT/${TAG}_pmc_$name.txt"; }
pass wait1 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
pass wait2 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VALU
pass wait3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass wait4 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass wait5 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE
