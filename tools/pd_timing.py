"""PlausibleDiversity / pairwise_distance_chamfer call time (chamfer.py:20-59,130-195): B x P poses, 500 model points."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from tests import helpers as H
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
for B, P in ((30, 30), (100, 100)):
    T_est = H.random_rigid(B, seed=1, trans=0.05).cuda()
    T_p = H.random_rigid(P, seed=2, trans=0.05).cuda()
    pd = pv.PlausibleDiversity(obj, num_model_points_eval=500)
    for bidir in (False, True):
        pd(torch.linalg.inv(T_est), T_p, bidirectional=bidir); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r = pd(torch.linalg.inv(T_est), T_p, bidirectional=bidir)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"B={B} P={P} bidirectional={bidir}: {dt*1e3:.2f} ms per evaluation ({B*P*500*(2 if bidir else 1)/dt:.2e} point-to-mesh distances/s)  plausibility {float(r.plausibility):.3f} coverage {float(r.coverage):.3f}")
