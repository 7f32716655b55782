"""C4 (RobotSDF, 200 configurations x 262,144 random points, 100 KB link grids) a few times, for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import workloads as Wk
robot = Wk.build_c4(0.02, 0.1)
A, P = 200, 1 << 18
robot.set_joint_configuration(Wk.c4_joint_configs(A))
pts = Wk.c4_points(P)
val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
for _ in range(3):
    robot.query_into(pts, val, grad)
torch.cuda.synchronize()
print(val.shape, float(val.mean()))
