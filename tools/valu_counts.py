"""rocprofv3 --pmc counter CSVs of tools/run_valu.py -> per-CALL totals of the kernels that belong to the query (every
launch of a call summed: the mesh paths launch several kernels per call), as JSON for bench.py's VALU rooflines.
usage: valu_counts.py <key> <calls> <workload text> <command text> <kernel regex> <csv>..."""
import csv, json, re, sys, collections
key, calls, workload, command, pattern = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], re.compile(sys.argv[5])
tot, per_kernel = collections.defaultdict(float), collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[6:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if pattern.search(name) and "mesh_prepare" not in name:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            per_kernel[re.sub(r"\(.*", "", name)[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: v / calls for k, v in tot.items()}
if out.get("SQ_INSTS_VALU"):
    out["active_lanes"] = out.get("SQ_THREAD_CYCLES_VALU", 0.0) / out["SQ_INSTS_VALU"] if out.get("SQ_THREAD_CYCLES_VALU") else None
out.update({"workload": workload, "command": command, "calls_profiled": calls,
            "kernels": {k: {c: v / calls for c, v in d.items()} for k, d in per_kernel.items()}})
print(json.dumps({key: out}))
