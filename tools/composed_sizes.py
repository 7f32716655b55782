"""ComposedSDF of 8 drills (A = 1) across query sizes: where the one-point-per-lane kernel hands over to the wave-tile one."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import workloads as Wk
from bench_configs import gpu_time
cached = Wk.build_c2_cache()
comp = Wk.build_c3(cached)
out = []
for P in (100_000, 400_000, 400_003, 1 << 20, (1 << 20) + 3, 1 << 21, (1 << 21) + 3, 1 << 22, (1 << 22) + 3, 1 << 23, (1 << 23) + 3):
    pts = Wk.c3_points(P, seed=P)
    val = torch.empty((1, P), dtype=torch.float32, device="cuda"); grad = torch.empty((1, P, 3), dtype=torch.float32, device="cuda")
    t, _ = gpu_time(lambda: comp.query_into(pts, val, grad), reps=20)
    out.append(f"P={P}: {t*1e6:.1f} us")
print(" | ".join(out))
