"""Per-launch means of rocprofv3 --pmc counters for one kernel, launches grouped in runs of N (tools/run_cq64.py: 3 per margin).
usage: pmc_table.py <counter_collection.csv> <kernel substring> <launches per group> [group labels...]"""
import csv, sys, collections
path, kernel, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
labels = sys.argv[4:]
rows = collections.defaultdict(dict)  # dispatch id -> counter -> value
for r in csv.DictReader(open(path)):
    if kernel in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
groups = [ids[i:i + n] for i in range(0, len(ids), n)]
names = sorted({c for d in rows.values() for c in d})
print("| counter | " + " | ".join(labels[i] if i < len(labels) else f"group {i}" for i in range(len(groups))) + " |")
print("|---|" + "---|" * len(groups))
for c in names:
    print(f"| {c} | " + " | ".join(f"{sum(rows[i].get(c, 0.0) for i in g) / len(g):.4g}" for g in groups) + " |")
