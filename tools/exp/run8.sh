cd $GRAFT_REPO_ROOT
bash tools/collect_benches.sh r03 > gpurun_out/collect_benches_r03.txt 2>&1
bash tools/collect_profiles.sh r03 > gpurun_out/collect_profiles_r03.txt 2>&1
