"""C4 through RobotSDF.__call__ (allocation included): direct vs bucketed path, both grid sizes."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import workloads as Wk
from bench_configs import gpu_time
for padding in (0.1, 1.0):
    robot = Wk.build_c4(0.02, padding)
    A, P = 200, 1 << 18
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    out = {}
    for mode in (False, True):
        robot.sdf.bucket_points = mode
        t, _ = gpu_time(lambda: robot(pts), reps=8)
        out[mode] = (t, robot(pts))
    same = torch.equal(out[False][1][0], out[True][1][0]) and torch.equal(out[False][1][1].nan_to_num(3.), out[True][1][1].nan_to_num(3.))
    robot.sdf.bucket_points = "auto"
    print(f"padding {padding}: direct {out[False][0]*1e3:.3f} ms | bucketed {out[True][0]*1e3:.3f} ms | identical bits {same} | auto picks bucketed: {robot.sdf._bucketing_pays(A, P)}")
