"""Does already-ordered input still want the Morton sort of the bucketed path?  README-size link grids (padding 1.0), A = 200,
P = 262,144 points given (a) uniformly random, (b) as a 512 x 512 slice in grid order (z fastest), (c) as a 64^3 grid in grid
order: direct call (bucket_points = False) against the bucketed call (True), and what "auto" picks."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch
import pytorch_volumetric_amd as pv, workloads as Wk
from mesh_probe import gpu_ms
for padding in (1.0, 0.1):
    robot = Wk.build_c4(resolution=0.02, padding=padding)
    robot.set_joint_configuration(Wk.c4_joint_configs(200))
    lo, hi = np.array(Wk.ARM_BOX[0]), np.array(Wk.ARM_BOX[1])
    ax = [torch.linspace(float(lo[d]), float(hi[d]), 512) for d in range(3)]
    slice_pts = torch.cartesian_prod(ax[0], torch.tensor([0.02]), ax[2]).cuda()
    ax64 = [torch.linspace(float(lo[d]), float(hi[d]), 64) for d in range(3)]
    cube_pts = torch.cartesian_prod(*ax64).cuda()
    for name, pts in (("uniform random", Wk.c4_points(1 << 18)), ("512 x 512 slice, grid order", slice_pts), ("64^3 grid, grid order", cube_pts)):
        out = []
        for mode in (False, True, "auto"):
            robot.sdf.bucket_points = mode
            out.append(gpu_ms(lambda: robot(pts), reps=10)[0])
        robot.sdf.bucket_points = "auto"
        picks = robot.sdf._bucketing_pays(200, pts.shape[0], pts)
        print(f"padding {padding}, {name}: direct %.3f ms | bucketed %.3f ms | auto %.3f ms (auto sorts: {picks})" % tuple(out))
