cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "two_round" -s 2>&1 | grep "two rounds\|passed\|failed\|Error" | tail -25 > gpurun_out/r03_tests7.log
