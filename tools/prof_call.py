import sys, os, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from tests import helpers as H
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("d", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
pts = H.uniform_points(1000, [-0.2] * 3, [0.3] * 3, seed=1).cuda()
for _ in range(100): cached(pts)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): cached(pts)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
