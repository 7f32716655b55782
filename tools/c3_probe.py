"""Round 6: C3 (and C4 without scratch) through pvamd_composed_query: fused in-workgroup grouping against the round-5 kernels."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from grouped_probe import graph_time

cached = Wk.build_c2_cache()
comp = Wk.build_c3(cached)
for P in (1 << 22, 1 << 20, 1 << 18):
    pts = Wk.c3_points(P)
    val = torch.empty((1, P), device="cuda"); grad = torch.empty((1, P, 3), device="cuda")
    out = {}
    for mode in (False, "auto"):
        comp.group_points = mode
        out[mode] = graph_time(lambda: comp.query_into(pts, val, grad))
    print(f"C3 P {P}: round-5 kernels {out[False]:.4f} ms | entry point's choice {out['auto']:.4f} ms", flush=True)
robot = Wk.build_c4(0.02, 0.1)
A, P = 200, 1 << 18
robot.set_joint_configuration(Wk.c4_joint_configs(A))
pts = Wk.c4_points(P)
val = torch.empty((A, P), device="cuda"); grad = torch.empty((A, P, 3), device="cuda")
lib = _lib.load()
grids = robot.sdf._leaf_grids(pts.device); tfd = robot.sdf._tf_device(pts.device)
for name, flags in (("round-5 wave-tile", _lib.COMPOSED_NO_GROUPING), ("fused (sort per configuration)", 0)):
    t = graph_time(lambda: lib.pvamd_composed_query(_lib.ptr(grids), 8, _lib.ptr(tfd), A, _lib.ptr(pts), P, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr()))
    print(f"C4 via pvamd_composed_query, {name}: {t:.4f} ms", flush=True)
robot.sdf.group_points = "auto"
t = graph_time(lambda: robot.sdf.query_into(pts, val, grad))
print(f"C4 via query_into (pre-pass + grouped): {t:.4f} ms")
