"""Broad/narrow-phase counters of the mesh kernels (needs libpvamd built with -DPVAMD_MESH_STATS:
tools/build_variant.sh stats pytorch_volumetric_amd/csrc/mesh.hip -DPVAMD_MESH_STATS; PVAMD_LIB=tools/variants/libpvamd_stats.so).
Counters are per wave (summed over the waves of a point group); "per-block" = per 64 points."""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import mesh_io, _lib
import workloads as H

lib = _lib.load()
NAMES = ["tile_passes (64 tile spheres each)", "tiles_visited", "record_passes (64 records at (c,Q))",
         "survivors (tested per point)", "closest_pairs", "ray_pairs", "drains", "survivors_any_near (rect per point)",
         "enqueues", "group tests per point", "-", "-", "-", "-", "-", "-", "cycles in drain_closest", "cycles in visit_tile (all)",
         "cycles in survivor loops (incl. their drains)", "cycles in scan_mesh (main launch)", "cycles in seed + greedy",
         "cycles in parts_of_group", "points needing a group of the pass (summed over record passes)"]


def stats(reset=True):
    buf = (ctypes.c_ulonglong * 32)()
    torch.cuda.synchronize()
    lib.pvamd_debug_stats(buf, 1 if reset else 0)
    return list(buf)


def report(tag, P, F):
    st = stats()
    print(f"== {tag}: P={P} F={F} blocks={-(-P // 64)} tiles={-(-F // 256)}")
    for n, v in zip(NAMES, st):
        print(f"   {n:40s} {v:14d}  per-block {v / (-(-P // 64)):10.1f}")
    print(f"   closest pairs / point = {st[4] / P:.2f}   ray pairs / point = {st[5] / P:.2f}   "
          f"pairs per drain = {(st[4] + st[5]) / max(1, st[6]):.1f}")


def main():
    m = mesh_io.uv_sphere_mesh(0.1, 250, 200)
    sphere = pv.MeshObjectFactory(mesh=m)
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 19)
    src = H.uniform_points(n, [-0.15] * 3, [0.15] * 3, seed=2).cuda()
    W = torch.eye(4).unsqueeze(0).cuda()
    pv.batch_chamfer_dist(W, src, sphere, scale=1000.0)
    stats()
    pv.batch_chamfer_dist(W, src, sphere, scale=1000.0)
    report("C5 chamfer sphere (1/4 of the points)", n, m.faces.shape[0])

    drill = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
    stats()
    c = pv.CachedSDF("drill", 0.002, drill.bounding_box(padding=0.05), pv.MeshSDF(drill), device="cuda", cache_path=None)
    report("cache build drill 0.002 pad 0.05", int(np.prod(c.voxels.shape)), 15728)

    pts = H.uniform_points(10000, [-0.2] * 3, [0.2] * 3, seed=1).cuda()
    stats()
    pv.MeshSDF(drill)(pts)
    report("10k random points, drill", 10000, 15728)


if __name__ == "__main__":
    main()
