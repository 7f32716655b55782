"""Driver for rocprofv3 --pmc passes over the 64M-point C2 launch: 3 launches each at margin 0.05 (52 % of the points out
of range -- the bench's large_batch), -0.001 (every point gathers) and 9 (no point gathers), in that order."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, workloads as Wk
cached = Wk.build_c2_cache()
P = 1 << 26
val = torch.empty((P,), dtype=torch.float32, device="cuda"); grad = torch.empty((P, 3), dtype=torch.float32, device="cuda")
for margin in (0.05, -0.001, 9.0):
    pts = Wk.c2_points(cached, P, seed=99, margin=margin)
    for _ in range(3):
        cached.query_into(pts, val, grad)
    torch.cuda.synchronize()
    del pts
