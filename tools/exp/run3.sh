cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scan_mapping or nan_cotangents" 2>&1 | tail -15 > gpurun_out/r03_tests2.log
python tools/exp/scan_bench.py > gpurun_out/r03_scan_bench.json 2> gpurun_out/r03_scan_bench.err
