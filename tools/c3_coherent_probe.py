"""C3 on spatially ORDERED inputs (a regular grid in C order; Hilbert-sorted random points): does the entry point's in-workgroup
regrouping cost anything where the caller's order is already coherent?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from grouped_probe import graph_time
cached = Wk.build_c2_cache()
comp = Wk.build_c3(cached)
ax = torch.linspace(-0.5, 0.5, 161)
grid = torch.cartesian_prod(ax, ax, ax).cuda().contiguous()
rnd = Wk.c3_points(grid.shape[0])
order = _lib.morton_order(rnd).long()
hil = rnd[order].contiguous()
slab = torch.cartesian_prod(torch.linspace(-0.5, 0.5, 2048), torch.linspace(-0.5, 0.5, 2048), torch.tensor([0.1])).cuda().contiguous()
for name, pts in (("random", rnd), ("161^3 grid, C order", grid), ("Hilbert-sorted random", hil), ("2048^2 planar slice", slab)):
    P = pts.shape[0]
    val = torch.empty((1, P), device="cuda"); grad = torch.empty((1, P, 3), device="cuda")
    out = {}
    for mode in (False, "auto"):
        comp.group_points = mode
        out[mode] = graph_time(lambda: comp.query_into(pts, val, grad))
    print(f"C3 {name} ({P} points): round-5 kernels {out[False]:.4f} ms | entry point's choice {out['auto']:.4f} ms", flush=True)
print("-- threshold: random points, the entry point's choice vs forced in-workgroup regrouping vs none")
comp.group_points = "auto"
for P in (1 << 19, 1 << 20, 3 << 19, 1 << 21, 3 << 20, 1 << 22):
    pts = Wk.c3_points(P)
    val = torch.empty((1, P), device="cuda"); grad = torch.empty((1, P, 3), device="cuda")
    out = {}
    for name, flags in (("none", _lib.COMPOSED_NO_GROUPING), ("forced", _lib.COMPOSED_FORCE_FUSED), ("auto", 0)):
        comp._leaf_grids(pts.device); comp._query_flags = flags
        out[name] = graph_time(lambda: comp.query_into(pts, val, grad))
    print(f"C3 random P {P} ({P // 4096} chunks): none {out['none']:.4f} | forced {out['forced']:.4f} | auto {out['auto']:.4f} ms", flush=True)
