"""C4 (or C3) through query_into N times with the given kernel flags -- the workload of tools/pmc_composed.sh.
usage: run_composed.py c4|c3 <flags> <calls>"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import workloads as Wk
which, flags, calls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if which == "c4":
    sdf = Wk.build_c4(0.02, 0.1)
    A, P = 200, 1 << 18
    sdf.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    comp = sdf.sdf
else:
    sdf = comp = Wk.build_c3(Wk.build_c2_cache())
    A, P = 1, 1 << 22
    pts = Wk.c3_points(P)
val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
comp._leaf_grids(pts.device)
comp._query_flags |= flags
for _ in range(calls):
    sdf.query_into(pts, val, grad)
torch.cuda.synchronize()
