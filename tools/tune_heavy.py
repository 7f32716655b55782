"""C5 chamfer through the Python call over (parts, waves per block, x-blocks) of the heavy-groups launch (-DPVAMD_MESH_TUNE build)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as H
from ab_mesh import timed
mesh = H.build_c5_mesh(); pts = H.c5_points(1 << 21); W = torch.eye(4).unsqueeze(0).cuda()
run = lambda: pv.batch_chamfer_dist(W, pts, mesh, scale=1000.0)
print("default %.3f" % timed(run, 5), flush=True)
for xb in (512, 1024, 2048):
    for waves in (4, 2):
        row = []
        for parts in (16, 32, 48, 64, 98, 130, 195):
            os.environ.update(PVAMD_TUNE_HPARTS=str(parts), PVAMD_TUNE_HWAVES=str(waves), PVAMD_TUNE_HBLOCKS=str(xb))
            row.append("%d %.3f" % (parts, timed(run, 4)))
        print("x-blocks %d, %d waves: " % (xb, waves) + " | ".join(row), flush=True)
