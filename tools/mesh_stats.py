"""Broad/narrow-phase counters of the mesh kernels (needs libpvamd built with -DPVAMD_MESH_STATS)."""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import mesh_io, _lib
from tests import helpers as H

lib = _lib.load()
NAMES = ["tile_steps(block)", "tiles_loaded(block)", "tiles_scanned(wave)", "tri_sphere_tests(wave)", "closest_pairs",
         "ray_pairs", "drains(wave)", "tris_any_closest(wave)", "tris_any_ray(wave)", "group_tests(wave)", "rect_tests(wave)"]


def stats(reset=True):
    buf = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize()
    lib.pvamd_debug_stats(buf, 1 if reset else 0)
    return list(buf)


def report(tag, P, F):
    st = stats()
    print(f"== {tag}: P={P} F={F} blocks={-(-P // 64)} tiles={-(-F // 256)}")
    for n, v in zip(NAMES, st):
        print(f"   {n:28s} {v:14d}  per-block {v / (-(-P // 64)):10.1f}")
    print(f"   closest pairs / point = {st[4] / P:.2f}   ray pairs / point = {st[5] / P:.2f}   "
          f"lane fill of any-closest tris = {st[4] / max(1, st[7]) / 64:.3f}  any-ray = {st[5] / max(1, st[8]) / 64:.3f}")


# C5
m = mesh_io.uv_sphere_mesh(0.1, 250, 200)
sphere = pv.MeshObjectFactory(mesh=m)
n = 1 << 19
src = H.uniform_points(n, [-0.15] * 3, [0.15] * 3, seed=2).cuda()
W = torch.eye(4).unsqueeze(0).cuda()
pv.batch_chamfer_dist(W, src, sphere, scale=1000.0)
stats()
pv.batch_chamfer_dist(W, src, sphere, scale=1000.0)
report("C5 chamfer sphere (1/4 points)", n, m.faces.shape[0])

# cache build drill
drill = pv.MeshObjectFactory(os.path.join("tests", "golden", "meshes", "ycb_power_drill.npz"))
stats()
c = pv.CachedSDF("drill", 0.002, drill.bounding_box(padding=0.05), pv.MeshSDF(drill), clean_cache=True,
                 cache_path="/tmp/mesh_stats_cache.pkl")
report("cache build drill 0.002 pad 0.05", int(np.prod(c.voxels.shape)), 15728)

# cache build wrench
wrench = pv.MeshObjectFactory(os.path.join("tests", "golden", "meshes", "offset_wrench_nogrip.obj"))
stats()
c = pv.CachedSDF("wrench", 0.002, wrench.bounding_box(padding=0.05), pv.MeshSDF(wrench), clean_cache=True,
                 cache_path="/tmp/mesh_stats_cache.pkl")
report("cache build wrench 0.002 pad 0.05", int(np.prod(c.voxels.shape)), 1263)

# C1
pts = H.uniform_points(10000, [-0.2] * 3, [0.2] * 3, seed=1).cuda()
stats()
pv.MeshSDF(drill)(pts)
report("C1 10k pts drill", 10000, 15728)
