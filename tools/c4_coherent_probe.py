"""C4 (200 configurations, 100 KB link grids) on ordered inputs: the chunk-grouped pair against the round-5 wave-tile kernel."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from grouped_probe import graph_time
robot = Wk.build_c4(0.02, 0.1)
A = 200
robot.set_joint_configuration(Wk.c4_joint_configs(A))
rnd = Wk.c4_points(1 << 18)
hil = rnd[_lib.morton_order(rnd).long()].contiguous()
sl = torch.cartesian_prod(torch.linspace(-0.7, 0.7, 512), torch.tensor([0.02]), torch.linspace(-0.2, 1.5, 512)).cuda().contiguous()
g3 = torch.cartesian_prod(torch.linspace(-0.7, 0.7, 64), torch.linspace(-0.7, 0.7, 64), torch.linspace(-0.2, 1.5, 64)).cuda().contiguous()
for name, pts in (("random", rnd), ("Hilbert-sorted random", hil), ("512x512 planar slice", sl), ("64^3 grid, C order", g3)):
    P = pts.shape[0]
    val = torch.empty((A, P), device="cuda"); grad = torch.empty((A, P, 3), device="cuda")
    out = {}
    for mode in (False, True):
        robot.sdf.group_points = mode
        out[mode] = graph_time(lambda: robot.sdf.query_into(pts, val, grad))
    print(f"C4 {name} ({P} points): round-5 kernel {out[False]:.4f} ms | pre-pass + grouped {out[True]:.4f} ms", flush=True)
