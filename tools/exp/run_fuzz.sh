# a long randomised session over all fuzzers with fresh seeds (round 3): tools/exp/run_fuzz.sh <cases> <seed>
cd $GRAFT_REPO_ROOT
C=${1:-300}; SEED=${2:-1003}
for f in lists depth_order project raster sequence render refine fused; do
  echo "== fuzz_$f ($C cases, seed $SEED)"
  timeout 1500 python tools/exp/fuzz_$f.py $C $SEED 2>&1 | grep -v "amdgpu.ids" | tail -4
done > gpurun_out/r03_fuzz_long.txt 2>&1
