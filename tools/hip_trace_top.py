"""Longest API calls of a rocprofv3 --hip-trace / --hsa-trace csv directory (no kernel data): name, duration, start."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        try:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        except (KeyError, ValueError):
            continue
        rows.append((e - s, r.get("Function") or r.get("Name") or "?", s, r.get("Domain", ""), f.rsplit("/", 1)[-1]))
if not rows:
    print("no api trace rows found under", sys.argv[1]); sys.exit(0)
t0 = min(r[2] for r in rows)
print(f"{len(rows)} API calls traced; the 40 longest (ms, function, seconds since the first traced call, domain):")
for d, name, s, dom, f in sorted(rows, reverse=True)[:40]:
    print(f"  {d / 1e6:10.3f} ms  {name:44s} t={(s - t0) / 1e9:8.3f} s  {dom}")
# nesting: which long HSA calls sit inside which long HIP calls
longs = [r for r in rows if r[0] > 2e6]
print(f"calls above 2 ms: {len(longs)}")
