"""Work counters of the C5 mesh for points near the sphere's centre only (every group is handed over): what a heavy group costs.
PVAMD_LIB=tools/variants/libpvamd_stats.so python tools/heavy_stats.py [radius_m]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
import workloads as H
from mesh_stats import stats, report
from ab_mesh import timed
R = float(sys.argv[1]) if len(sys.argv) > 1 else 0.025
mesh = H.build_c5_mesh()
g = torch.Generator().manual_seed(0)
n = 745 * 64
d = torch.randn(n, 3, generator=g); d = d / d.norm(dim=1, keepdim=True) * (torch.rand(n, 1, generator=g) ** (1 / 3)) * R
pts = d.float().cuda()
W = torch.eye(4).unsqueeze(0).cuda()
run = lambda: pv.batch_chamfer_dist(W, pts, mesh, scale=1000.0)
run(); stats(); run()
report(f"{n} points within {R} m of the centre of the C5 sphere", n, mesh.num_faces)
print("ms per call %.3f" % timed(run, 5))
