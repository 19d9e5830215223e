#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/r03 && tar -xzf tools/r04/r03_src.tgz -C /tmp/r03 && make -C /tmp/r03/gaussian-splatting-toolkit_amd/csrc -j16 > /tmp/r03/build.log 2>&1
cat > /tmp/nogc.py <<'PY'
import gc, runpy, sys, os
gc.disable()
root = sys.argv[1]
sys.argv = [os.path.join(root, "tools", "render_bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
for i in 1 2; do
  echo "r03            : $(cd /tmp/r03 && python tools/render_bench.py 2>/dev/null | tail -1 | cut -c64-100)"
  echo "r03 runpy nogc : $(python /tmp/nogc.py /tmp/r03 2>/dev/null | tail -1 | cut -c64-100)"
  echo "now            : $(python tools/render_bench.py 2>/dev/null | tail -1 | cut -c64-100)"
  echo "now runpy nogc : $(python /tmp/nogc.py $GRAFT_REPO_ROOT 2>/dev/null | tail -1 | cut -c64-100)"
  echo "now 1000 frames: $(python tools/render_bench.py --frames 1000 2>/dev/null | tail -1 | cut -c64-100)"
  echo "r03 1000 frames: $(cd /tmp/r03 && python tools/render_bench.py --frames 1000 2>/dev/null | tail -1 | cut -c64-100)"
done
