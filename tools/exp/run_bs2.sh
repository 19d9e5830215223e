#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/one.py <<'P'
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "gaussian-splatting-toolkit_amd"))
import rasterizer.cuda as C
n = int(sys.argv[1]); rng = np.random.default_rng(0)
kind = os.environ.get("DIST", "uniform")
d = rng.uniform(2.5, 7.5, n) if kind == "uniform" else np.abs(rng.normal(5.0, 0.7, n)) + 0.2
d = torch.from_numpy(d.astype(np.float32)).cuda()
r = np.ones(n, np.int32); r[rng.integers(0, n, n // 10)] = 0
r = torch.from_numpy(r).cuda()
for _ in range(20):
    C.depth_order(d, r, None)
torch.cuda.synchronize()
P
for n in ${SIZES:-1000000 3000000}; do
rm -rf /tmp/prof; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /tmp/one.py $n > /tmp/prof.log 2>&1; tail -3 /tmp/prof.log
python - <<P
import csv, glob
f = glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)[0]
print("n = $n")
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
P
done
