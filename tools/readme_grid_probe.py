"""C4 with README-size link grids (8 x 21 MB): direct, bucketed (global Hilbert sort + un-permute), chunk-grouped (inline-exact)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from grouped_probe import graph_time
robot = Wk.build_c4(0.02, 1.0)
lib = _lib.load()
for A, P in ((200, 1 << 18), (20, 1 << 18), (200, 1 << 16)):
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    comp = robot.sdf
    dev = pts.device
    grids = comp._leaf_grids(dev); tfd = comp._tf_device(dev); flags = comp._query_flags
    val = torch.empty((A, P), device=dev); grad = torch.empty((A, P, 3), device=dev)
    t_direct = graph_time(lambda: lib.pvamd_composed_query(_lib.ptr(grids), 8, _lib.ptr(tfd), A, _lib.ptr(pts), P, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr()), reps=5)
    ref = (val.clone(), grad.clone())
    comp.bucket_points = True
    t_b = graph_time(lambda: robot(pts), reps=5)
    scratch = _lib.group_points(pts)
    def grouped():
        lib.pvamd_group_points(_lib.ptr(pts), P, _lib.ptr(scratch), _lib.stream_ptr())
        lib.pvamd_composed_query_grouped(_lib.ptr(grids), 8, _lib.ptr(tfd), A, _lib.ptr(scratch), P, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr())
    val.zero_(); grad.zero_()
    t_g = graph_time(grouped, reps=5)
    same = torch.equal(ref[0].view(torch.int32), val.view(torch.int32)) and torch.equal(ref[1].view(torch.int32), grad.view(torch.int32))
    print(f"README-size grids A {A} P {P} flags {flags}: direct {t_direct:.3f} ms | bucketed robot(points) {t_b:.3f} ms | chunk-grouped {t_g:.3f} ms | same bits {same}", flush=True)
    comp.bucket_points = "auto"
