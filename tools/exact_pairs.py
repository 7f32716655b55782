"""Exact (narrow-phase) point-triangle pairs the mesh kernels actually evaluate on C1 and C5 -- needs the stats build:
  tools/build_variant.sh stats pytorch_volumetric_amd/csrc/mesh.hip -DPVAMD_MESH_STATS
  PVAMD_LIB=tools/variants/libpvamd_stats.so python tools/exact_pairs.py > profiles/r03_exact_pairs.json
tools/bench_configs.py turns these into "flop of the exact tests / time" (the honest work rate of a culling kernel; the
brute-force-equivalent pair count says what a plain double loop would have had to do, not what was done)."""
import sys, os, ctypes, json
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk

lib = _lib.load()


def stats():
    buf = (ctypes.c_ulonglong * 32)()
    torch.cuda.synchronize()
    lib.pvamd_debug_stats(buf, 1)
    return list(buf)


out = {}
drill = Wk.build_drill()
sdf = pv.MeshSDF(drill)
_, grid_pts = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(0.01))
pts = grid_pts[torch.randperm(len(grid_pts), generator=torch.Generator().manual_seed(0))[:10_000]].cuda()
sdf(pts); stats(); sdf(pts)
st = stats()
out["C1"] = {"points": 10_000, "triangles": drill.num_faces, "closest_pairs": st[4], "ray_pairs": st[5],
             "survivors_tested_per_point": st[3]}
mesh = Wk.build_c5_mesh()
src = Wk.c5_points(1 << 21)
W = torch.eye(4).unsqueeze(0).cuda()
pv.batch_chamfer_dist(W, src, obj_factory=mesh, scale=1000.0); stats(); pv.batch_chamfer_dist(W, src, obj_factory=mesh, scale=1000.0)
st = stats()
out["C5"] = {"points": 1 << 21, "triangles": mesh.num_faces, "closest_pairs": st[4], "ray_pairs": st[5],
             "survivors_tested_per_point": st[3]}
out["source"] = "PVAMD_LIB=tools/variants/libpvamd_stats.so python tools/exact_pairs.py (mesh.hip built with -DPVAMD_MESH_STATS)"
print(json.dumps(out, indent=1))
