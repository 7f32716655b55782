"""README-size link grids, per-lane composed kernel (P < 32,768): with and without the leaves' pooled levels (ComposedSDF.pooled_leaves)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import pytorch_volumetric_amd as pv
import workloads as Wk
from grouped_probe import graph_time
_, pts = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
pts = pts.cuda()
robot = Wk.build_c4(0.02, 1.0)
for A in (20, 200):
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    for name, p in (("slice 15,251", pts), ("random 15,251", Wk.c4_points(15251)), ("random 4,096", Wk.c4_points(4096)), ("random 30,000", Wk.c4_points(30000))):
        val = torch.empty((A, p.shape[0]), device="cuda"); grad = torch.empty((A, p.shape[0], 3), device="cuda")
        out = []
        for pooled in (False, True, False, True):
            robot.sdf.pooled_leaves = pooled
            robot.sdf.query_into(p, val, grad)
            out.append(graph_time(lambda: robot.sdf.query_into(p, val, grad)))
        print(f"A {A} {name}: plain {out[0]:.4f} / {out[2]:.4f} ms | pooled {out[1]:.4f} / {out[3]:.4f} ms", flush=True)
