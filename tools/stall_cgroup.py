"""Is the 20-40 ms host stall of a call loop the container's CPU quota?  cgroup cpu.stat (nr_throttled / throttled_usec) and the
per-thread scheduler statistics of the main thread before and after the loops of tools/stall_trace.py, under the default
OpenMP wait policy and with OMP_WAIT_POLICY=passive / one torch thread."""
import os, subprocess, sys, time


def cpu_stat():
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(path):
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
    return out


def quota():
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(path):
            return open(path).read().strip()
    return "?"


if len(sys.argv) > 1 and sys.argv[1] == "child":
    a = cpu_stat()
    sched0 = open("/proc/self/schedstat").read().split()
    t0 = time.perf_counter()
    sys.argv = ["stall_trace.py"]
    exec(open(os.path.join(os.path.dirname(__file__), "stall_trace.py")).read())
    b = cpu_stat()
    sched1 = open("/proc/self/schedstat").read().split()
    print("cgroup cpu.stat delta:", {k: b[k] - a.get(k, 0) for k in b if b[k] != a.get(k, 0)})
    print(f"main thread: on-cpu {(int(sched1[0]) - int(sched0[0])) / 1e6:.0f} ms, runnable-but-waiting {(int(sched1[1]) - int(sched0[1])) / 1e6:.1f} ms, "
          f"wall {(time.perf_counter() - t0) * 1e3:.0f} ms, threads now {len(os.listdir('/proc/self/task'))}")
else:
    print("cpu quota (cpu.max):", quota(), " cpus visible:", os.cpu_count(), " affinity:", len(os.sched_getaffinity(0)))
    for label, env in (("default", {}), ("OMP_WAIT_POLICY=passive", {"OMP_WAIT_POLICY": "passive"}),
                       ("OMP_NUM_THREADS=1", {"OMP_NUM_THREADS": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        keep = [l for l in r.stdout.splitlines() if l.startswith(("steps above", "cgroup", "main thread"))]
        print(f"== {label}\n   " + "\n   ".join(l[:600] for l in keep))
