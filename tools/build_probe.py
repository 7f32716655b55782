"""Round 6: wall / event time of CachedSDF(...) for the three bench grids, fused build (pvamd_cache_build) vs the generic one."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
import workloads as Wk
from bench_legs import CACHE_BUILDS


class Generic(pv.MeshSDF):
    pass


for key, mesh_name, res, pad, _ in CACHE_BUILDS:
    obj = pv.MeshObjectFactory(Wk.mesh_path(mesh_name))
    for label, gt in (("fused", pv.MeshSDF(obj)), ("generic", Generic(obj))):
        gt(torch.zeros(64, 3).cuda())
        wall, ev = [], []
        for i in range(8):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            c = pv.CachedSDF(key, res, obj.bounding_box(padding=pad), gt, device="cuda", cache_path=None)
            e1.record(); torch.cuda.synchronize()
            if i >= 2:
                wall.append((time.perf_counter() - t0) * 1e3); ev.append(e0.elapsed_time(e1))
        print(f"{key:14s} {label:8s} wall {np.median(wall):.3f} ms | events {np.median(ev):.3f} ms | host part {np.median(wall) - np.median(ev):.3f}", flush=True)
