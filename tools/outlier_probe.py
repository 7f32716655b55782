"""Round 2's tools/latency.py read 192 us for ComposedSDF(8)(pts) at P = 15,251 against 20 us at P = 1,000 and 100,000
(profiles/r02_probes.txt).  Reproduce: per-call wall times (sorted), the same through query_into, neighbouring counts."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_volumetric_amd as pv
from tests import helpers as H
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("d", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
comp = pv.ComposedSDF([cached] * 8, pv.Transform3d(matrix=H.random_rigid(8, seed=0)))


def per_call(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.sort(np.array(ts))
    return f"median {ts[n // 2]:.1f} p90 {ts[int(n * .9)]:.1f} max {ts[-1]:.1f}"


def pipelined(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


for P in (1000, 15000, 15251, 15252, 15360, 16384, 30000, 60000, 100000):
    pts = H.uniform_points(P, [-0.2] * 3, [0.3] * 3, seed=1).cuda()
    val = torch.empty((1, P), device="cuda"); grad = torch.empty((1, P, 3), device="cuda")
    print(f"P={P}: comp(pts) pipelined {pipelined(lambda: comp(pts)):.1f} us, per call [{per_call(lambda: comp(pts))}] | "
          f"query_into pipelined {pipelined(lambda: comp.query_into(pts, val, grad)):.1f} us, per call [{per_call(lambda: comp.query_into(pts, val, grad))}] | "
          f"cached(pts) pipelined {pipelined(lambda: cached(pts)):.1f} us")
