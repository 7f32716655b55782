"""Where in a rocprofv3 kernel trace of bench.py do the slow 1M-point launches sit?  (round 6: the trace's average for the headline kernel is
0.5-0.7 us above the HIP-event figure of the same run.)  Usage: python tools/kt_sequence.py <dir with *kernel_trace.csv>
Classes by the gap between a launch's start and the end of the dispatch before it:
  queued    < 1 us   the next node of a hipGraph replay was waiting in the queue: its 'start' is the previous 'end', so the duration the
                     trace reports is the whole launch-to-launch period (what HIP events around the graph / K measure)
  paced     1-20 us  eager calls, the host issuing them slower than the GPU runs them (the drop-in call loop, warm-up steps)
  isolated  > 20 us  after a synchronize"""
import csv, glob, sys, json
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cls = {"queued": [], "paced": [], "isolated": []}
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "cached_query_direct<true, false, 2, 16>" in r["Kernel_Name"] and (r.get("Grid_Size_X") or r.get("Grid_Size")) == "524288":
        gap = (s - prev_end) / 1e3 if prev_end else 1e9
        cls["queued" if gap < 1.0 else ("paced" if gap < 20.0 else "isolated")].append((e - s) / 1e3)
    prev_end = e
out = {}
for k, v in cls.items():
    v.sort()
    if v:
        out[k] = {"calls": len(v), "avg_us": round(sum(v) / len(v), 3), "p10_us": v[len(v) // 10], "median_us": v[len(v) // 2], "p90_us": v[len(v) * 9 // 10]}
allv = sorted(d for v in cls.values() for d in v)
out["all"] = {"calls": len(allv), "avg_us": round(sum(allv) / len(allv), 3), "median_us": allv[len(allv) // 2]}
print(json.dumps(out, indent=1))
