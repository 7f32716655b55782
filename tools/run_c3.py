"""C3 (ComposedSDF, 8 drills, 4M random points) a few times, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as H
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("drill", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
comp = pv.ComposedSDF([cached] * 8, pv.Transform3d(matrix=H.random_rigid(8, seed=0)))
pts = H.uniform_points(1 << 22, [-0.5] * 3, [0.5] * 3, seed=0).cuda()
for _ in range(3):
    comp(pts)
torch.cuda.synchronize()
