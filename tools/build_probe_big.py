"""Fused (pvamd_cache_build, 4x4x4-brick order) against the generic construction (Hilbert sort) on a 17 M-voxel grid."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
import workloads as Wk


class Generic(pv.MeshSDF):
    pass


obj = pv.MeshObjectFactory(Wk.mesh_path("ycb_power_drill.npz"))
for res, pad in ((0.001, 0.05), (0.002, 0.05)):
    for label, gt in (("fused", pv.MeshSDF(obj)), ("generic", Generic(obj))):
        gt(torch.zeros(64, 3).cuda())
        wall = []
        for i in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c = pv.CachedSDF("big", res, obj.bounding_box(padding=pad), gt, device="cuda", cache_path=None)
            torch.cuda.synchronize()
            if i:
                wall.append((time.perf_counter() - t0) * 1e3)
        print(f"drill res {res} pad {pad} {tuple(c._view.shape)} {label:8s} {np.median(wall):.3f} ms", flush=True)
        if label == "fused":
            ref = c._packed.clone()
        else:
            print("   same bits:", torch.equal(ref.view(torch.int32), c._packed.view(torch.int32)))
        del c
