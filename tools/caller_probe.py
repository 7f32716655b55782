"""Call times of the hot path's other callers (one MI355X): pairwise chamfer against a CachedSDF (pvamd_chamfer_grid), a RobotSDF
whose leaves are MeshSDFs (the reference's default link_sdf_cls: the per-leaf path), CachedSDF with the LOOKUP_GT_SDF strategy."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch
import pytorch_volumetric_amd as pv, workloads as Wk
from mesh_probe import gpu_ms
from tests import helpers as H

obj = Wk.build_drill()
cached = Wk.build_c2_cache(obj)
pts, _, _ = pv.sample_mesh_points(obj, name="drill", num_points=500, dbpath=None)
T = H.random_rigid(100, seed=1, trans=0.05).cuda()
Tp = H.random_rigid(100, seed=2, trans=0.05).cuda()
W = torch.einsum("bij,pjk->bpik", torch.linalg.inv(T), Tp).reshape(-1, 4, 4)
print("batch_chamfer_dist, 10,000 transforms x 500 points: vs CachedSDF %.3f ms | vs the mesh %.3f ms" %
      (gpu_ms(lambda: pv.batch_chamfer_dist(W, pts, obj_sdf=cached))[0], gpu_ms(lambda: pv.batch_chamfer_dist(W, pts, obj))[0]))
with tempfile.TemporaryDirectory() as tmp:
    chain = Wk.synthetic_arm(tmp)
    robot = pv.RobotSDF(chain, path_prefix=tmp)  # MeshSDF leaves
    _, slice_pts = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
    slice_pts = slice_pts.cuda()
    for A in (1, 20):
        robot.set_joint_configuration(Wk.c4_joint_configs(A) if A > 1 else None)
        print(f"RobotSDF over 8 MeshSDF leaves (per-leaf path), A={A} x M={slice_pts.shape[0]}: %.3f ms" % gpu_ms(lambda: robot(slice_pts), reps=5)[0])
lk = pv.CachedSDF("drill", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), out_of_bounds_strategy=pv.OutOfBoundsStrategy.LOOKUP_GT_SDF,
                  device="cuda", cache_path=None)
q = Wk.c2_points(lk, 1 << 20, seed=3)
print("CachedSDF LOOKUP_GT_SDF, 1M points (52 %% out of range -> mesh query on 550k points): %.3f ms" % gpu_ms(lambda: lk(q), reps=5)[0])
