#!/bin/bash
# L2 (TCC) and HBM-side counters of the composed kernel on C4 with README-size link grids (padding 1.0), random and
# Morton-sorted points: rocprofv3 --pmc passes -> gpurun_out/$1/
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
cat > /tmp/run_readme.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import workloads as Wk
from pytorch_volumetric_amd import _lib
robot = Wk.build_c4(0.02, 1.0)
A, P = 200, 1 << 18
robot.set_joint_configuration(Wk.c4_joint_configs(A))
pts = Wk.c4_points(P)
spts = pts[_lib.morton_order(pts).long()].contiguous()
val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
for q in (pts, spts, pts, spts):
    robot.query_into(q, val, grad)
torch.cuda.synchronize()
PY
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/tcc -o c4 --output-format csv -- python /tmp/run_readme.py > $O/tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o c4 --output-format csv -- python /tmp/run_readme.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o c4 --output-format csv -- python /tmp/run_readme.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o c4 --output-format csv -- python /tmp/run_readme.py > $O/kt.log 2>&1
python - <<PY > $O/summary.txt
import csv, glob
for d in ("tcc", "fetch", "write"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "composed_query_wave" in r["Kernel_Name"]]
        names = sorted({r["Counter_Name"] for r in rows})
        for n in names:
            vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == n]
            print(d, n, "per launch (random, sorted, random, sorted):", " ".join("%.4g" % v for v in vals))
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "composed" in r["Name"]:
            print("kernel-trace", r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
find $O -name "*.csv" -size +1M -delete
cat $O/summary.txt
