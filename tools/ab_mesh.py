"""Mesh-kernel timings for A/B builds: PVAMD_LIB=tools/variants/libpvamd_X.so python tools/ab_mesh.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import mesh_io
import workloads as H


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    out = []
    sphere = pv.MeshObjectFactory(mesh=mesh_io.uv_sphere_mesh(0.1, 250, 200))
    src = H.uniform_points(1 << 21, [-0.15] * 3, [0.15] * 3, seed=2).cuda()
    W = torch.eye(4).unsqueeze(0).cuda()
    out.append("C5 %.2f ms" % timed(lambda: pv.batch_chamfer_dist(W, src, sphere, scale=1000.0), 4))
    drill = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
    gt = pv.MeshSDF(drill)
    _, grid = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(padding=0.05))
    grid = grid.cuda()
    out.append("drill grid %d pts %.2f ms" % (grid.shape[0], timed(lambda: gt(grid), 4)))
    pts = H.uniform_points(10000, [-0.2] * 3, [0.2] * 3, seed=1).cuda()
    out.append("C1-like 10k random %.3f ms" % timed(lambda: gt(pts), 10))
    big = H.uniform_points(1 << 20, [-0.2] * 3, [0.3] * 3, seed=3).cuda()
    out.append("1M random box %.2f ms" % timed(lambda: gt(big), 3))
    surf, _, _ = pv.sample_mesh_points(drill, num_points=1 << 21, seed=0, dbpath=None, device="cuda")
    surf = (surf + 0.001 * torch.randn_like(surf)).float()
    out.append("2M near-surface chamfer %.2f ms" % timed(lambda: pv.batch_chamfer_dist(W, surf, drill), 3))
    for n in (3_000, 10_000, 30_000, 60_000, 100_000, 200_000):
        q = H.uniform_points(n, [-0.2] * 3, [0.3] * 3, seed=n).cuda()
        out.append("%dk random %.3f ms" % (n // 1000, timed(lambda: gt(q), 6)))
    print(os.environ.get("PVAMD_LIB", "default"), " | ".join(out))


if __name__ == "__main__":
    main()
