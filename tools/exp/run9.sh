cd $GRAFT_REPO_ROOT
python -m pytest tests/ -q -m gpu -v > gpurun_out/r03_full_gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1
