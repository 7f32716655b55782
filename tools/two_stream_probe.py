"""Do K back-to-back C2 launches finish sooner when issued as two independent chains (two streams captured into one graph,
separate output buffers) than as one chain?  Kernel boundaries cost ~1.5 us on one stream."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import workloads as Wk
cached = Wk.build_c2_cache()
P = 1 << 20
pts = Wk.c2_points(cached, P, seed=1)
outs = [(torch.empty((P,), device="cuda"), torch.empty((P, 3), device="cuda")) for _ in range(4)]
for _ in range(3000): cached.query_into(pts, *outs[0])
torch.cuda.synchronize()
def graph(K, chains):
    main = torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in range(chains - 1)]
    main.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            for s in sides: s.wait_stream(main)
            for i in range(K):
                c = i % chains
                st = main if c == 0 else sides[c - 1]
                with torch.cuda.stream(st):
                    cached.query_into(pts, *outs[c])
            for s in sides: main.wait_stream(s)
    torch.cuda.current_stream().wait_stream(main)
    return g
for K in (20, 2000):
    for chains in (1, 2, 4):
        g = graph(K, chains)
        for _ in range(5): g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(7):
            done = torch.cuda.Event(); torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); done.record()
            while not done.query(): pass
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"K={K} chains={chains}: {best*1e6:.1f} us total, {best/K*1e6:.2f} us/step")
