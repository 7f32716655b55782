import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import mesh_io
import workloads as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 21)
m = mesh_io.uv_sphere_mesh(0.1, 250, 200)
sphere = pv.MeshObjectFactory(mesh=m)
src = H.uniform_points(n, [-0.15] * 3, [0.15] * 3, seed=2).cuda()
W = torch.eye(4).unsqueeze(0).cuda()
for _ in range(2):
    e = pv.batch_chamfer_dist(W, src, sphere, scale=1000.0)
torch.cuda.synchronize()
print(e)
