"""C4 with the shared point set pre-sorted along a Morton curve (outputs left in sorted order): how much does the composed
kernel gain from spatially coherent wave tiles?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from bench_configs import gpu_time
for padding in (0.1, 1.0):
    robot = Wk.build_c4(0.02, padding)
    A, P = 200, 1 << 18
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    order = _lib.morton_order(pts).long()
    spts = pts[order].contiguous()
    val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
    t_r, _ = gpu_time(lambda: robot.query_into(pts, val, grad), reps=10)
    t_s, _ = gpu_time(lambda: robot.query_into(spts, val, grad), reps=10)
    t_sort, _ = gpu_time(lambda: _lib.morton_order(pts), reps=10)
    print(f"padding {padding}: random {t_r*1e3:.3f} ms | morton-sorted {t_s*1e3:.3f} ms | sort itself {t_sort*1e3:.3f} ms")
