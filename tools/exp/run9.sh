cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1
