import os, sys
sys.path.insert(0, os.getcwd())
import torch, pytorch_volumetric_amd as pv, workloads as Wk
sys.path.insert(0, "tools")
from mesh_probe import gpu_ms
drill = Wk.build_drill(); sdf = pv.MeshSDF(drill)
_, grid_pts = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(0.01))
g = torch.Generator().manual_seed(0)
pts = grid_pts[torch.randperm(len(grid_pts), generator=g)[:10_000]].cuda()
bb = drill.bounding_box(padding=0.05)
print(os.environ.get("PVAMD_LIB", "product"), "C1 %.3f ms (min %.3f)" % gpu_ms(lambda: sdf(pts), reps=30), end=" | ")
for n in (1000, 3000, 30000, 60000, 100000):
    rnd = Wk.uniform_points_device(n, bb[:, 0], bb[:, 1], 5)
    print(f"{n}: %.3f (min %.3f)" % gpu_ms(lambda: sdf(rnd)), end=" | ")
print()
