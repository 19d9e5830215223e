cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -x -q -m gpu -k "two_round" 2>&1 | tail -8 > gpurun_out/r03_tests8.log
bash tools/exp/run17.sh
bash tools/exp/run18.sh
