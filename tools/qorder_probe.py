"""Does a processing order along the points' CLOSEST SURFACE POINTS beat the order along the points themselves?
(the kernels take any permutation: object_frame_closest_point(points, order=...))"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as H
from ab_mesh import timed
drill = H.build_drill(); sphere = H.build_c5_mesh()
_, grid_pts = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(0.01))
cases = [("drill C1 grid 10k", drill, grid_pts[torch.randperm(len(grid_pts), generator=torch.Generator().manual_seed(0))[:10_000]].cuda())]
for n in (3000, 10_000, 30_000, 100_000, 400_000, 2_000_000):
    cases.append((f"drill random {n}", drill, H.uniform_points(n, [-0.2] * 3, [0.3] * 3, seed=n).cuda()))
for n in (10_000, 100_000, 2_097_152):
    cases.append((f"sphere random {n}", sphere, H.uniform_points(n, [-0.15] * 3, [0.15] * 3, seed=2).cuda()))
for name, obj, pts in cases:
    res = obj.object_frame_closest_point(pts)
    o_p = _lib.morton_order(pts, min_points=0)
    o_q = _lib.morton_order(res.closest.float().contiguous(), min_points=0)
    t_p = timed(lambda: obj.object_frame_closest_point(pts, order=o_p), 8)
    t_q = timed(lambda: obj.object_frame_closest_point(pts, order=o_q), 8)
    r2 = obj.object_frame_closest_point(pts, order=o_q)
    same = torch.equal(res.distance, r2.distance) and torch.equal(res.closest, r2.closest)
    print(f"{name}: order along the points {t_p:.3f} ms | along their closest surface points {t_q:.3f} ms | same bits {same}", flush=True)
# heavy groups only: points within 2.5 cm of the sphere's centre (every group is handed over)
g = torch.Generator().manual_seed(0)
n = 745 * 64
d = torch.randn(n, 3, generator=g); d = d / d.norm(dim=1, keepdim=True) * (torch.rand(n, 1, generator=g) ** (1 / 3)) * 0.025
pts = d.float().cuda()
res = sphere.object_frame_closest_point(pts)
o_p = _lib.morton_order(pts, min_points=0)
o_q = _lib.morton_order(res.closest.float().contiguous(), min_points=0)
dirs = (pts / pts.norm(dim=1, keepdim=True)).contiguous()
o_d = _lib.morton_order(dirs, min_points=0)
for nm, o in (("points", o_p), ("closest surface points", o_q), ("directions from the centroid", o_d)):
    t = timed(lambda: sphere.object_frame_closest_point(pts, order=o), 5)
    print(f"centre ball {n}: order along the {nm} {t:.3f} ms", flush=True)
