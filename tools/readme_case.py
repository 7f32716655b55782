"""The reference README's own RobotSDF timing case (README.md:177-200): A joint configurations x M = 15,251 query points
(the 0.01 m grid over [-1, 0.5] x [-0.5, 0.5] x [-0.2, 0.8] subsampled as the README does), link caches at resolution 0.02
with padding 1.0 (README.md:150-151) and with padding 0.1, on the synthetic 7-DOF arm (the KUKA assets are not in this tree).
Published: 37.7 ms (A = 20) and 128.6 ms (A = 200) on an RTX 2080 Ti."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import torch, pytorch_volumetric_amd as pv, workloads as Wk
from mesh_probe import gpu_ms
M = 15251
pts = Wk.uniform_points(M, [-1.0, -0.5, -0.2], [0.5, 0.5, 0.8], seed=7).cuda()
for padding in (1.0, 0.1):
    robot = Wk.build_c4(resolution=0.02, padding=padding)
    for A in (20, 200):
        th = Wk.c4_joint_configs(A)
        robot.set_joint_configuration(th)
        q = gpu_ms(lambda: robot(pts), reps=20)
        s = gpu_ms(lambda: robot.set_joint_configuration(th), reps=20)
        print(f"padding {padding}: A={A} x M={M}: query %.3f ms (min %.3f) = %.3g (config, point) pairs/s | set_joint_configuration %.3f ms"
              % (q[0], q[1], A * M / (q[0] * 1e-3), s[0]))
