import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_volumetric_amd as pv
from tests import helpers as H
obj = pv.MeshObjectFactory(H.mesh_path("offset_wrench_nogrip.obj"))
gt = pv.MeshSDF(obj)
_, pts = pv.get_coordinates_and_points_in_grid(0.001, pv.get_divisible_range_by_resolution(0.001, obj.bounding_box(padding=0.05)))
print(len(pts), obj.bounding_box())
pts = pts.cuda()
for n in (65536, 1 << 20, len(pts)):
    sub = pts[:n].contiguous()
    gt(sub); torch.cuda.synchronize()
    t0 = time.perf_counter(); v, g = gt(sub); torch.cuda.synchronize(); print(n, "points:", (time.perf_counter() - t0) * 1e3, "ms")
obj._mesh_desc()
tiles = obj._tiles_dev.cpu().numpy()[: (obj.num_faces + 255) // 256]
rec = obj._rec_dev.cpu().numpy()
print("tiles", tiles)
print("tri r min/mean/max", rec[:, 3].min(), rec[:, 3].mean(), rec[:, 3].max())
