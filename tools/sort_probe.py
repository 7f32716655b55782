"""How long does a library radix sort (torch.sort -> rocPRIM) of (key, index) pairs take beside pvamd_morton_order?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from pytorch_volumetric_amd import _lib
from ab_mesh import timed
for n in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22):
    pts = torch.rand(n, 3, device="cuda")
    keys = torch.randint(0, 1 << 21, (n,), device="cuda", dtype=torch.int32)
    t_lib = timed(lambda: torch.sort(keys), 10)
    t_own = timed(lambda: _lib.morton_order(pts), 10)
    print(f"{n}: torch.sort int32 keys+indices {t_lib:.3f} ms | pvamd_morton_order {t_own:.3f} ms", flush=True)
