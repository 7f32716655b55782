"""Few-points mesh query: time per call over (points, parts, waves per block) with a -DPVAMD_MESH_TUNE build
(PVAMD_LIB=tools/variants/libpvamd_tune.so; PVAMD_TUNE_PARTS / PVAMD_TUNE_WAVES are read by pvamd_mesh_query per call)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as H
from ab_mesh import timed
which = sys.argv[1] if len(sys.argv) > 1 else "drill"
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz")) if which == "drill" else H.build_c5_mesh()
gt = pv.MeshSDF(obj)
box = ([-0.2] * 3, [0.3] * 3) if which == "drill" else ([-0.15] * 3, [0.15] * 3)
SIZES = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1000, 3000, 10_000, 20_000, 40_000, 70_000, 100_000, 131_000)
PARTS = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (0, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128)
for n in SIZES:
    q = H.uniform_points(n, box[0], box[1], seed=n).cuda()
    row = []
    for waves in (4, 2):  # (one wave per block was 0.139 ms on C1 against 0.104: no longer instantiated)
        for parts in PARTS:
            if parts == 0 and waves != 4:
                continue
            os.environ["PVAMD_TUNE_PARTS"] = str(parts); os.environ["PVAMD_TUNE_WAVES"] = str(waves)
            row.append("%dx%d %.3f" % (waves, parts, timed(lambda: gt(q), 8)))
    del os.environ["PVAMD_TUNE_PARTS"]
    row.append("auto %.3f" % timed(lambda: gt(q), 8))
    print(which, n, " | ".join(row), flush=True)
