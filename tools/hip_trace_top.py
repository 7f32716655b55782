"""Longest HIP API calls of a rocprofv3 --hip-trace --hsa-trace --kernel-trace csv directory, each with the HSA calls nested
inside it (same thread, contained in time) and the kernel it dispatched (matched by correlation id): names the runtime call
behind a host stall.  usage: hip_trace_top.py <dir> [min ms]"""
import csv, glob, sys, collections
d = sys.argv[1]
floor = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6


def rows(pattern):
    for f in glob.glob(d + "/**/" + pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            yield r


hip = [r for r in rows("*hip_api_trace.csv")]
hsa = [r for r in rows("*hsa_api_trace.csv")]
kern_rows = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows("*kernel_trace.csv")))
kern = {}
if not hip:
    print("no hip api rows under", d); sys.exit(0)
t0 = min(int(r["Start_Timestamp"]) for r in hip)
print(f"{len(hip)} HIP calls, {len(hsa)} HSA calls, {len(kern)} kernel dispatches traced")
by_thread = collections.defaultdict(list)
for r in hsa:
    by_thread[r["Thread_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
long_calls = sorted((r for r in hip if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > floor and "Synchronize" not in r["Function"]),
                    key=lambda r: int(r["Start_Timestamp"]))
for r in long_calls:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"\n{(e - s) / 1e6:9.3f} ms  {r['Function']}  at t = {(s - t0) / 1e9:.3f} s, thread {r['Thread_Id']}"
          f"  -> kernel: {kern.get(r.get('Correlation_Id'), '(none)')[:110]}")
    import bisect
    i = bisect.bisect_left(kern_rows, (s, ""))
    print("      kernels dispatched before / after its start:", " | ".join(k[:60] for _, k in kern_rows[max(0, i - 2):i]), " >>> ",
          " | ".join(f"{k[:70]} (+{(t - s) / 1e6:.2f} ms)" for t, k in kern_rows[i:i + 3]))
    inner = [(b - a, n, a) for (a, b, n) in by_thread.get(r["Thread_Id"], []) if a >= s and b <= e]
    agg = collections.defaultdict(lambda: [0, 0])
    for dur, n, _ in inner:
        agg[n][0] += dur; agg[n][1] += 1
    for n, (tot, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
        print(f"      {tot / 1e6:9.3f} ms in {cnt:4d} x {n}")
    print(f"      HSA time inside: {sum(x[0] for x in inner) / 1e6:.3f} ms of {(e - s) / 1e6:.3f} ms")
