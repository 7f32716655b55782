"""The reference README's own RobotSDF timing case (README.md:177-200): A joint configurations x M = 15,251 query points --
the README's points themselves, `get_coordinates_and_points_in_grid(0.01, [[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]])`, an
ordered 151 x 1 x 101 slice -- link caches at resolution 0.02 with padding 1.0 (README.md:150-151) and with padding 0.1,
on the synthetic 7-DOF arm (the KUKA assets are not in this tree, so this is NOT like for like).
Published: 37.7 ms (A = 20) and 128.6 ms (A = 200) on an RTX 2080 Ti.  Uniform random points of the same count and box are
timed beside the slice, and each case with the entry point's own kernel choice, the one-point-per-lane kernel (flag 2) and
the wave-tile kernel (flag 4)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np
import torch, pytorch_volumetric_amd as pv, workloads as Wk
from mesh_probe import gpu_ms

_, slice_pts = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
M = slice_pts.shape[0]
assert M == 15251
slice_pts = slice_pts.cuda()
rand_pts = Wk.uniform_points(M, [-1.0, -0.5, -0.2], [0.5, 0.5, 0.8], seed=7).cuda()
for padding in (1.0, 0.1):
    robot = Wk.build_c4(resolution=0.02, padding=padding)
    for A in (20, 200):
        th = Wk.c4_joint_configs(A)
        robot.set_joint_configuration(th)
        s = gpu_ms(lambda: robot.set_joint_configuration(th), reps=20)
        for name, pts in (("README slice", slice_pts), ("uniform random", rand_pts)):
            q = gpu_ms(lambda: robot(pts), reps=30)
            val = torch.empty((A, M), device="cuda"); grad = torch.empty((A, M, 3), device="cuda")
            comp = robot.sdf
            comp._leaf_grids(pts.device)
            base = comp._query_flags
            k = []
            for fl in (base, base | 2, base | 2 | 8, base | 4):
                comp._query_flags = fl
                k.append(gpu_ms(lambda: comp.query_into(pts, val, grad), reps=30)[0])
            comp._query_flags = base
            print(f"padding {padding}: A={A} x M={M} {name}: robot(pts) %.4f ms (min %.4f) = %.3g (config, point) pairs/s | "
                  f"kernel only: auto %.4f, per-lane %.4f, per-lane config-fastest %.4f, wave-tile %.4f ms | set_joint_configuration %.3f ms | "
                  f"published (RTX 2080 Ti, KUKA): {37.688577 if A == 20 else 128.645445:.1f} ms"
                  % (q[0], q[1], A * M / (q[0] * 1e-3), k[0], k[1], k[2], k[3], s[0]))
