"""C1 (MeshSDF on the drill, 10k points) a few times, for rocprofv3 --kernel-trace."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as H
drill = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
sdf = pv.MeshSDF(drill)
pts = H.uniform_points(10000, [-0.2] * 3, [0.2] * 3, seed=1).cuda()
for _ in range(5):
    sdf(pts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    v, g = sdf(pts)
torch.cuda.synchronize()
print("C1 wall per call: %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
