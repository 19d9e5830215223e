#!/bin/bash
# per-kernel times of the binning stages: tools/exp/binprof.sh <n> [env...]   (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
N=${1:-1000000}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/binprof
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/binprof -o bp -- python $ROOT/tools/exp/binbench.py $N 10 2>/dev/null | grep -v "^\[" | tail -3
python $ROOT/tools/summarize_prof.py /tmp/binprof /tmp/binprof/summary.json | grep -v "at::native\|rocclr" | head -${TOP:-14}
