"""README case (A = 20 / 200, 15,251 slice points, 21 MB and 100 KB link grids) and small-batch composed calls: ms per launch."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import pytorch_volumetric_amd as pv
import workloads as Wk
from grouped_probe import graph_time
_, pts = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
pts = pts.cuda()
rnd = Wk.c4_points(15251)
for pad in (0.1, 1.0):
    robot = Wk.build_c4(0.02, pad)
    for A in (20, 200):
        robot.set_joint_configuration(Wk.c4_joint_configs(A))
        val = torch.empty((A, pts.shape[0]), device="cuda"); grad = torch.empty((A, pts.shape[0], 3), device="cuda")
        t1 = graph_time(lambda: robot.sdf.query_into(pts, val, grad))
        t2 = graph_time(lambda: robot.sdf.query_into(rnd, val, grad))
        print(f"padding {pad} A {A}: slice {t1:.4f} ms | random 15,251 points {t2:.4f} ms", flush=True)
cached = Wk.build_c2_cache(); comp = Wk.build_c3(cached)
for P in (1 << 16, 1 << 18, 1 << 20):
    p = Wk.c3_points(P); v = torch.empty((1, P), device="cuda"); g = torch.empty((1, P, 3), device="cuda")
    print(f"C3-like P {P}: {graph_time(lambda: comp.query_into(p, v, g)):.4f} ms", flush=True)
