#!/bin/bash
export TMPDIR=/tmp
for v in "" tools/variants/libpvamd_inr8.so tools/variants/libpvamd_inr4.so; do
echo "== $v"
rm -rf /tmp/kt; PVAMD_LIB=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o c -- python tools/run_composed.py c4 0 12 > /tmp/kt.log 2>&1
python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "composed" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[2:]
    print(k, "n=%d mean %.1f us min %.1f" % (len(v), sum(v) / len(v), min(v)))
PY
done
timeout 600 python tools/composed_ab.py c4 2>&1 | grep "^C4"
timeout 600 python -m pytest tests/test_composed_queue_gpu.py -q -m gpu 2>&1 | tail -1
