"""One BASELINE workload CALLS times, for a rocprofv3 pass (tools/valu_session.sh): c1 | c3 | c4 | c5 | bd1 | bd2 | bw (cache builds)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk
which, calls = sys.argv[1], int(sys.argv[2])
if which == "c1":  # MeshSDF on the drill, 10k of the 0.002 m grid points (tests/test_sdf.py:46-48 of the reference)
    drill = Wk.build_drill()
    sdf = pv.MeshSDF(drill)
    _, grid_pts = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(0.01))
    pts = grid_pts[torch.randperm(len(grid_pts), generator=torch.Generator().manual_seed(0))[:10_000]].cuda()
    fn = lambda: sdf(pts)
elif which == "c3":
    comp = Wk.build_c3(Wk.build_c2_cache())
    P = 1 << 22
    pts = Wk.c3_points(P)
    val = torch.empty((1, P), dtype=torch.float32, device="cuda"); grad = torch.empty((1, P, 3), dtype=torch.float32, device="cuda")
    fn = lambda: comp.query_into(pts, val, grad)
elif which == "c4":
    robot = Wk.build_c4(0.02, 0.1)
    A, P = 200, 1 << 18
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
    fn = lambda: robot.query_into(pts, val, grad)
elif which in ("bd1", "bd2", "bw"):  # bench.py CACHE_BUILDS: one CachedSDF construction per call
    mesh_name, res, pad = {"bd1": ("ycb_power_drill.npz", 0.01, 0.1), "bd2": ("ycb_power_drill.npz", 0.002, 0.01),
                           "bw": ("offset_wrench_nogrip.obj", 0.001, 0.05)}[which]
    obj = pv.MeshObjectFactory(Wk.mesh_path(mesh_name))
    gt = pv.MeshSDF(obj)
    gt(torch.zeros(64, 3).cuda())
    fn = lambda: pv.CachedSDF(which, res, obj.bounding_box(padding=pad), gt, device="cuda", cache_path=None)
else:
    mesh = Wk.build_c5_mesh()
    pts = Wk.c5_points(1 << 21)
    W = torch.eye(4).unsqueeze(0).cuda()
    fn = lambda: pv.batch_chamfer_dist(W, pts, obj_factory=mesh, scale=1000.0)
import time
for _ in range(calls):
    fn()
    torch.cuda.synchronize()
    time.sleep(0.01)  # a gap in the kernel trace between calls (tools/valu_session.py groups launches by it)
