"""wave-tile kernel vs one-point-per-lane kernel (flags bit 1) on C3 / C4 / README-size C4, random and ordered points."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import workloads as Wk
from pytorch_volumetric_amd import _lib
from bench_configs import gpu_time
def run(name, sdf, pts, A):
    P = pts.shape[0]
    val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
    comp = getattr(sdf, "sdf", sdf)
    comp._leaf_grids(pts.device)
    base = comp._query_flags
    out = []
    for fl in (0, 1, 2):
        comp._query_flags = fl
        t, _ = gpu_time(lambda: sdf.query_into(pts, val, grad), reps=10)
        out.append(t)
    comp._query_flags = base
    print(f"{name}: wave-tile/estimate {out[0]*1e3:.4f} ms | wave-tile/inline {out[1]*1e3:.4f} ms | one-point-per-lane {out[2]*1e3:.4f} ms")
cached = Wk.build_c2_cache()
comp = Wk.build_c3(cached)
p3 = Wk.c3_points(1 << 22)
run("C3 random", comp, p3, 1)
run("C3 sorted", comp, p3[_lib.morton_order(p3).long()].contiguous(), 1)
for padding in (0.1, 1.0):
    robot = Wk.build_c4(0.02, padding)
    robot.set_joint_configuration(Wk.c4_joint_configs(200))
    p4 = Wk.c4_points(1 << 18)
    run(f"C4 pad {padding} random", robot, p4, 200)
    run(f"C4 pad {padding} sorted", robot, p4[_lib.morton_order(p4).long()].contiguous(), 200)
