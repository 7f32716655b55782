"""Mesh-kernel throughput for points NEAR the surface (the chamfer / plausible-pose regime): 2M surface samples + noise."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
from tests import helpers as H
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
pts, _, _ = pv.sample_mesh_points(obj, num_points=1 << 21, seed=0, dbpath=None, device="cuda")
W = torch.eye(4).unsqueeze(0).cuda()
for noise in (0.0, 0.001, 0.01):
    q = (pts + noise * torch.randn_like(pts)).float()
    pv.batch_chamfer_dist(W, q, obj); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); err = pv.batch_chamfer_dist(W, q, obj); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"noise {noise*1e3:.0f} mm: {ms:.2f} ms for {len(q)} points ({len(q)/ms*1e3:.2e} pts/s), chamfer {float(err[0]):.4f} mm^2")
