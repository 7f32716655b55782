#!/bin/bash
# kernel trace of one MeshSDF call on 10k grid points of the drill (BASELINE C1) -> gpurun_out/$1/c1_kernels.txt
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/c1kt -o c1 --output-format csv -- python tools/run_c1.py > $O/c1kt.log 2>&1
python - <<PY > $O/c1_kernels.txt
import csv, glob
for f in glob.glob("$O/c1kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print("%-90s calls %4s avg us %9.2f total us %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
cat $O/c1_kernels.txt
