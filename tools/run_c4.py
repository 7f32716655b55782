import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
from tests import helpers as H
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench_configs import synthetic_arm
with tempfile.TemporaryDirectory() as tmp:
    chain = synthetic_arm(tmp)
    robot = pv.RobotSDF(chain, path_prefix=tmp, link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=0.1, device="cuda", cache_path=None))
A, P4 = 200, 1 << 18
th0 = torch.tensor([0.0, -np.pi / 4, 0.0, np.pi / 2, 0.0, np.pi / 4, 0.0])
th = torch.cat((th0.view(1, -1), th0 + torch.randn(A - 1, 7, generator=torch.Generator().manual_seed(0)) * 0.1))
robot.set_joint_configuration(th)
pts4 = H.uniform_points(P4, [-0.7, -0.7, -0.2], [0.7, 0.7, 1.5], seed=1).cuda()
for _ in range(3):
    v, g = robot(pts4)
torch.cuda.synchronize()
print(v.shape, float(v.mean()))
