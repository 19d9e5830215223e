#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1
cat > /tmp/prof_rb.py <<'PY'
import cProfile, io, pstats, sys, os, runpy
root = sys.argv[1]
sys.argv = [os.path.join(root, "tools", "render_bench.py")]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
PY
echo "=== r03"; python /tmp/prof_rb.py /tmp/r03 2>/dev/null | grep -v "^$" | head -34
echo "=== now"; GSR_SPECULATE=0 python /tmp/prof_rb.py $GRAFT_REPO_ROOT 2>/dev/null | grep -v "^$" | head -34
