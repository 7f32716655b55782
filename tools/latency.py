"""Python-level latency of small calls through the drop-in API (what an interactive / planner user sees)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_volumetric_amd as pv
from tests import helpers as H
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("d", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
comp = pv.ComposedSDF([cached] * 8, pv.Transform3d(matrix=H.random_rigid(8, seed=0)))
mesh = pv.MeshSDF(obj)
def lat(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for P in (1000, 15251, 100000):
    pts = H.uniform_points(P, [-0.2] * 3, [0.3] * 3, seed=1).cuda()
    print(f"P={P}: CachedSDF {lat(lambda: cached(pts)):.1f} us  ComposedSDF(8) {lat(lambda: comp(pts)):.1f} us  "
          f"MeshSDF {lat(lambda: mesh(pts), 20):.1f} us  outside_surface {lat(lambda: cached.outside_surface(pts)):.1f} us")
