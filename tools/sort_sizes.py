"""pvamd_morton_order over point counts (which sort serves which size: PVAMD_LIB / PVAMD_RADIX_FROM pick an A/B build)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from pytorch_volumetric_amd import _lib
from ab_mesh import timed
out = []
for n in (20_000, 1 << 15, 1 << 16, 100_000, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 1 << 21, 1 << 22, 1 << 24):
    pts = torch.rand(n, 3, device="cuda")
    for _ in range(3):
        _lib.morton_order(pts, min_points=0)
    t = min(timed(lambda: _lib.morton_order(pts, min_points=0), 20) for _ in range(3))
    t2 = min(timed(lambda: _lib.morton_order(pts, min_points=0, want_inverse=True, want_sorted=True), 20) for _ in range(3))
    out.append(f"{n}: {t:.4f} ms (order only) {t2:.4f} ms (order + inverse + sorted points)")
print(f"PVAMD_RADIX_FROM={os.environ.get('PVAMD_RADIX_FROM', 'default')}\n" + "\n".join(out), flush=True)
