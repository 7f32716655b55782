"""C3 (8 drills, 4M random points) through pvamd_composed_query for A/B builds: PVAMD_LIB=tools/variants/libpvamd_X.so python tools/c3_only_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk
from grouped_probe import graph_time

cached = Wk.build_c2_cache()
comp = Wk.build_c3(cached)
P = 1 << 22
pts = Wk.c3_points(P)
val = torch.empty((1, P), device="cuda"); grad = torch.empty((1, P, 3), device="cuda")
comp.group_points = "auto"
ts = [graph_time(lambda: comp.query_into(pts, val, grad)) for _ in range(3)]
print(os.environ.get("PVAMD_LIB", "product"), "C3 4M:", " ".join(f"{t:.4f}" for t in ts), flush=True)
