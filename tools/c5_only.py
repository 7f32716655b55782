"""C5 alone (for rocprofv3 --kernel-trace --stats): 12 chamfer calls, 2,097,152 points -> 99,500-triangle sphere."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk
mesh = Wk.build_c5_mesh()
pts = Wk.c5_points(1 << 21)
W = torch.eye(4).unsqueeze(0).cuda()
for _ in range(12):
    err = pv.batch_chamfer_dist(W, pts, obj_factory=mesh, scale=1000.0)
torch.cuda.synchronize()
print(float(err[0]))
