cd $GRAFT_REPO_ROOT
( cd tools/exp && ./atomic_bench 1000000 120 68 2 && ./atomic_bench 3000000 240 135 6 && ./atomic_bench 1000000 120 68 3 ) > gpurun_out/atomic_bench.txt 2>&1
python -m pytest tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_bench.py "tests/test_gpu_fullsize.py::test_config2_200k_sh3_1080p_forward_backward_vs_oracle" tests/test_gpu_dp.py "tests/test_gpu_kernels.py::test_nan_cotangents_at_undrawn_pixels_are_never_read" -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r03_tests1.log
bash tools/collect_attribution.sh r03
