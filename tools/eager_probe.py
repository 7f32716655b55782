import time, torch, sys
sys.path.insert(0, ".")
import workloads as Wk
cached = Wk.build_c2_cache()
for P in (15251, 1 << 20):
    pts = Wk.c2_points(cached, P, seed=3)
    val = torch.empty((P,), dtype=torch.float32, device="cuda"); grad = torch.empty((P, 3), dtype=torch.float32, device="cuda")
    for _ in range(200): cached.query_into(pts, val, grad)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(2000): cached.query_into(pts, val, grad)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 2000)
    bestc = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(2000): cached(pts)
        torch.cuda.synchronize()
        bestc = min(bestc, (time.perf_counter() - t0) / 2000)
    print(f"P {P}: query_into {best*1e6:.2f} us/call | cached(points) {bestc*1e6:.2f} us/call")
