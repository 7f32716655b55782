"""Where do the 5.8 us of the 1M-point C2 launch go?  The same launch with no point gathering (margin 9: every point takes the
bounding-box branch), every point gathering (margin -0.001), the bench mix; torch's copy of the same bytes; an empty launch --
each as a 2000-launch hipGraph, HIP events / 2000."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk


def graph_us(fn, n=2000, reps=5):
    for _ in range(50):
        fn()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


def main():
    cached = Wk.build_c2_cache()
    P = 1 << 20
    val = torch.empty((P,), dtype=torch.float32, device="cuda")
    grad = torch.empty((P, 3), dtype=torch.float32, device="cuda")
    out = []
    for label, margin in (("bench mix (52.7 % out of range)", 0.05), ("no point gathers (all out of range)", 9.0), ("every point gathers", -0.001)):
        pts = Wk.c2_points(cached, P, seed=99, margin=margin)
        out.append(f"{label}: {graph_us(lambda: cached.query_into(pts, val, grad)):.2f} us")
    tiny = Wk.c2_points(cached, 16384, seed=1)
    tv, tg = torch.empty((16384,), device="cuda"), torch.empty((16384, 3), device="cuda")
    out.append(f"16,384 points (same kernel, 4 workgroups): {graph_us(lambda: cached.query_into(tiny, tv, tg)):.2f} us")
    src = torch.empty(28 * P // 8, dtype=torch.float32, device="cuda"); dst = torch.empty_like(src)
    out.append(f"torch copy, 14 MB -> 14 MB: {graph_us(lambda: dst.copy_(src)):.2f} us")
    big = torch.empty(12 * P // 4, dtype=torch.float32, device="cuda")
    out.append(f"torch fill of 16 MB (write only): {graph_us(lambda: dst[:4 * P].fill_(1.0)):.2f} us")
    out.append(f"torch sum-free read proxy: mul_ in place on 12 MB (read + write 12 MB each): {graph_us(lambda: big.mul_(1.0)):.2f} us")
    print("\n".join(out), flush=True)


if __name__ == "__main__":
    main()
