#!/bin/bash
# The driver-shaped bench line + the rocprofv3 kernel trace of the SAME command + PMC traffic passes -> gpurun_out/$1/
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
[ -z "$ONLY_KT" ] && python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_detail_k20.json > $O/bench_k20.json 2> $O/bench_k20.err
rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --detail $O/bench_detail_kt.json > $O/kt.log 2>&1
[ -z "$ONLY_KT" ] && rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-legs --detail /tmp/d1.json > $O/fetch.log 2>&1
[ -z "$ONLY_KT" ] && rocprofv3 --pmc WRITE_SIZE -d $O/write -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-legs --detail /tmp/d2.json > $O/write.log 2>&1
python - <<PY > $O/kernel_stats.md
# per (kernel, launch geometry): the same template instantiation serves the 1M-point steps and the 8M / 64M-point legs
import csv, glob, collections
rows = collections.defaultdict(list)
ctx = collections.defaultdict(list)  # the 1M-point launches by what ran right before them: (gap to the previous dispatch's end in us, duration)
for f in glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True):
    prev_end = None
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"][:96]
        # pvamd_cached_query picks the kernel by the point count (csrc/cached.hip cq_kind): one instantiation per size class of the bench
        if "cached_query_direct<true, false, 2, 16>" in name:  # 1024-thread workgroups of 2048 points
            name += " [1,048,576-point launches]" if grid == "524288" else " [%d-point launches, every point in range: the all_in_range_batch leg]" % (int(grid) * 2)
        elif "cached_query_direct<true, false, 4, 4>" in name:
            name += " [8M-point launches of the mid_batch leg]"
        elif "cached_query_wave<true, false, true, true>" in name:
            name += " [64M-point launches of the large_batch leg]" if d < 420 else " [1e8-point launches of the p1e8_batch leg]"
        rows[(name, grid, wg)].append(d)
        if "1,048,576-point launches" in name:
            ctx[(name, grid, wg)].append(((int(r["Start_Timestamp"]) - prev_end) / 1e3 if prev_end is not None else 1e9, d))
        prev_end = int(r["End_Timestamp"])
total = sum(sum(v) for v in rows.values())
print("| kernel | grid (threads) x workgroup | calls | total us | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|---|")
for (name, grid, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("| \`%s\` | %s x %s | %d | %.1f | %.2f | %.2f | %.2f | %.2f |" % (name, grid, wg, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / total))
# machine-readable: the dominant kernel's 1,048,576-point launches (bench.py roofline.frac_rocprof)
import json
dom = [(k, v) for k, v in rows.items() if "1,048,576-point launches" in k[0]]
if dom:
    (name, grid, wg), v = max(dom, key=lambda kv: len(kv[1]))
    def dist(x):
        x = sorted(x)
        return {"calls": len(x), "avg_us": sum(x) / len(x), "p10_us": x[len(x) // 10], "median_us": x[len(x) // 2], "p90_us": x[len(x) * 9 // 10]} if x else None
    c = ctx[(name, grid, wg)]
    # by the gap between a launch's start and the end of the dispatch before it (tools/kt_sequence.py):
    #   queued   < 1 us: the next node of a hipGraph replay was already waiting -- its start IS the previous end, so the trace's duration is the
    #                    whole launch-to-launch period, the quantity HIP events around the graph / K measure (bench.py roofline.launch_us)
    #   paced  1-20 us: eager calls issued by the host more slowly than the GPU runs them (drop-in call loop, warm-up steps)
    #   isolated > 20 us: after a synchronize
    split = {"all": dist(v), "queued": dist([d for g, d in c if g < 1.0]), "paced": dist([d for g, d in c if 1.0 <= g < 20.0]),
             "isolated": dist([d for g, d in c if g >= 20.0])}
    json.dump({"command": "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline",
               "dominant": {"kernel": name, "points": 1048576, "calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v), "by_context": split},
               "all": [{"kernel": k[0], "grid": k[1], "workgroup": k[2], "calls": len(v), "avg_us": sum(v) / len(v)} for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]]},
              open("$O/kernel_stats.json", "w"), indent=1)
PY
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
[ -z "$ONLY_KT" ] && python tools/pmc_summary.py $F $W "cached_query_direct<true, false, 2, 16>" 1048576 > $O/traffic.json
find $O -name "*.csv" -size +1M -delete
tail -c 2500 $O/bench_k20.json | head -c 100 > /dev/null
head -12 $O/kernel_stats.md; cat $O/traffic.json
