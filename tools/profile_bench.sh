#!/bin/bash
# The driver-shaped bench line + the rocprofv3 kernel trace of the SAME command + PMC traffic passes -> gpurun_out/$1/
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-legs > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-legs > $O/write.log 2>&1
python - <<PY > $O/kernel_stats.md
import csv, glob
print("| kernel | calls | total us | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:30]:
        print("| \`%s\` | %s | %.1f | %.2f | %.2f | %.2f | %s |" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W cached_query_wave 1048576 > $O/traffic.json
find $O -name "*.csv" -size +1M -delete
tail -c 2500 $O/bench_k20.json | head -c 100 > /dev/null
head -12 $O/kernel_stats.md; cat $O/traffic.json
