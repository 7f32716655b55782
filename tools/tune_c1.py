"""The bench's C1 workload (10k of the 0.002 m grid points of the drill) over parts x waves (-DPVAMD_MESH_TUNE build)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk
from ab_mesh import timed
drill = Wk.build_drill(); sdf = pv.MeshSDF(drill)
_, grid_pts = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(0.01))
pts = grid_pts[torch.randperm(len(grid_pts), generator=torch.Generator().manual_seed(0))[:10_000]].cuda()
print("auto %.4f" % timed(lambda: sdf(pts), 30), flush=True)
for waves in (4, 2):
    row = []
    for parts in (12, 16, 21, 26, 31, 40, 48, 62):
        os.environ.update(PVAMD_TUNE_PARTS=str(parts), PVAMD_TUNE_WAVES=str(waves))
        row.append("%d %.4f" % (parts, timed(lambda: sdf(pts), 30)))
    print("%d waves: " % waves + " | ".join(row), flush=True)
del os.environ["PVAMD_TUNE_PARTS"]
print("auto %.4f" % timed(lambda: sdf(pts), 30), flush=True)
