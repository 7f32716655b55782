"""Where does the wall time of a short timed region go?  K launches of the C2 kernel as (a) one hipGraph replay,
(b) K eager C-ABI calls; host time until the launch call(s) return vs time until the GPU is done."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import workloads as Wk
cached = Wk.build_c2_cache()
P = 1 << 20
pts = Wk.c2_points(cached, P, seed=1)
val = torch.empty((P,), dtype=torch.float32, device="cuda"); grad = torch.empty((P, 3), dtype=torch.float32, device="cuda")
step = lambda: cached.query_into(pts, val, grad)
for _ in range(3000): step()
torch.cuda.synchronize()
def region(fn):
    done = torch.cuda.Event()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); done.record()
    while not done.query(): pass
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t0) * 1e6, (t3 - t0) * 1e6
for K in (20, 100):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(K): step()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5): g.replay()
    samples = [region(g.replay) for _ in range(9)]
    print("   graph samples (GPU done, us):", " ".join(f"{r[1]:.0f}" for r in samples))
    rg = min(samples, key=lambda r: r[1])
    def eager():
        for _ in range(K): step()
    samples = [region(eager) for _ in range(9)]
    print("   eager samples (GPU done, us):", " ".join(f"{r[1]:.0f}" for r in samples))
    re = min(samples, key=lambda r: r[1])
    print(f"K={K:5d} graph: launch returns {rg[0]:8.1f} us, GPU done {rg[1]:8.1f} us ({rg[1]/K:6.2f}/step), after sync {rg[2]:8.1f} | "
          f"eager: calls return {re[0]:8.1f} us, GPU done {re[1]:8.1f} us ({re[1]/K:6.2f}/step), after sync {re[2]:8.1f}")

# as bench.py does it: 150 back-to-back replays, synchronize, then ONE timed region
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(20): step()
torch.cuda.current_stream().wait_stream(side)
for trial in range(4):
    for _ in range(150): g.replay()
    torch.cuda.synchronize()
    r = region(g.replay)
    print(f"bench-like trial {trial}: launch returns {r[0]:.1f} us, GPU done {r[1]:.1f} us, after sync {r[2]:.1f} us")
    time.sleep(0.2)
    r = region(g.replay)
    print(f"   after 0.2 s idle: GPU done {r[1]:.1f} us")
