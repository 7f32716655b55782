import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_volumetric_amd as pv
from tests import helpers as H
for name, res, pad in (("ycb_power_drill.npz", 0.005, 0.01), ("offset_wrench_nogrip.obj", 0.001, 0.05), ("ycb_power_drill.npz", 0.002, 0.01), ("ycb_power_drill.npz", 0.001, 0.05)):
    obj = pv.MeshObjectFactory(H.mesh_path(name))
    gt = pv.MeshSDF(obj)
    gt(torch.zeros(64, 3).cuda()); torch.cuda.synchronize()
    dt = 1e9
    for _ in range(3):  # best of 3: the first build of a size also pays for the allocator growing
        t0 = time.perf_counter()
        c = pv.CachedSDF(name, res, obj.bounding_box(padding=pad), gt, device="cuda", cache_path=None)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    n = int(np.prod(c._view.shape))
    print(f"{name} res={res} pad={pad}: grid {c._view.shape} = {n} voxels x {obj.num_faces} tris = {n*obj.num_faces:.3e} pairs in {dt*1e3:.1f} ms -> {n*obj.num_faces/dt:.3e} bf-eq pairs/s, {n/dt:.3e} voxels/s")
