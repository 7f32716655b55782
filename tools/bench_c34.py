"""C3 + C4 only (composed kernel A/B)."""
import sys, os, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import pytorch_volumetric_amd as pv
from tests import helpers as H
from bench_configs import synthetic_arm, gpu_time
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("drill", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
comp = pv.ComposedSDF([cached] * 8, pv.Transform3d(matrix=H.random_rigid(8, seed=0)))
pts3 = H.uniform_points(1 << 22, [-0.5] * 3, [0.5] * 3, seed=0).cuda()
t3, _ = gpu_time(lambda: comp(pts3), reps=30)
with tempfile.TemporaryDirectory() as tmp:
    robot = pv.RobotSDF(synthetic_arm(tmp), path_prefix=tmp, link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=0.1, device="cuda", cache_path=None))
A, P4 = 200, 1 << 18
th0 = torch.tensor([0.0, -np.pi / 4, 0.0, np.pi / 2, 0.0, np.pi / 4, 0.0])
th = torch.cat((th0.view(1, -1), th0 + torch.randn(A - 1, 7, generator=torch.Generator().manual_seed(0)) * 0.1))
robot.set_joint_configuration(th)
pts4 = H.uniform_points(P4, [-0.7, -0.7, -0.2], [0.7, 0.7, 1.5], seed=1).cuda()
t4, _ = gpu_time(lambda: robot(pts4), reps=20)
print(f"C3 {t3*1e3:.4f} ms ({(1<<22)/t3:.3e} q/s)   C4 {t4*1e3:.4f} ms ({A*P4/t4:.3e} pairs/s)")
