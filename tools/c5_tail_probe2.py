"""How much of C5 is its heavy tail?  Time the chamfer kernel on the C5 points with the ones nearest the sphere's centre
(equidistant to most of the surface) removed."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import torch, pytorch_volumetric_amd as pv, workloads as Wk
from mesh_probe import gpu_ms
mesh = Wk.build_c5_mesh()
p5 = Wk.c5_points(2_000_000)
H = torch.eye(4).unsqueeze(0).cuda()
r = p5.norm(dim=1)
for r0 in (0.0, 0.01, 0.02, 0.03, 0.05):
    q = p5[r >= r0].contiguous()
    print(f"|p| >= {r0}: {q.shape[0]} points (%.2f%% removed)  %.3f ms (min %.3f)" % ((100 - 100 * q.shape[0] / p5.shape[0],) + gpu_ms(lambda: pv.batch_chamfer_dist(H, q, mesh), reps=5)))
q = p5[(r - 0.1).abs() < 0.01].contiguous()
print(f"within 1 cm of the surface: {q.shape[0]} points  %.3f ms (min %.3f)" % gpu_ms(lambda: pv.batch_chamfer_dist(H, q, mesh), reps=5))
