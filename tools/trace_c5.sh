#!/bin/bash
# kernel trace of the C5 chamfer call (tools/run_c5.py) -> gpurun_out/$1/c5_kernels.txt
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/c5kt -o c5 --output-format csv -- python tools/run_c5.py > $O/c5kt.log 2>&1
python - <<PY > $O/c5_kernels.txt
import csv, glob
for f in glob.glob("$O/c5kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("%-90s calls %4s avg us %9.2f total us %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
cat $O/c5_kernels.txt
