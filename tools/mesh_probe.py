"""Timings of the mesh kernels on the workloads DESIGN.md quotes: C1 (10k grid points on the drill, whole call and kernel
only), the drill cache build at 0.002 m, 100k / 2M random points on the drill, C5 (chamfer, 2M points -> 99,500 triangles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk


def gpu_ms(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    drill = Wk.build_drill()
    sdf = pv.MeshSDF(drill)
    bb = drill.bounding_box(padding=0.05)
    _, grid_pts = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(0.01))
    g = torch.Generator().manual_seed(0)
    pts = grid_pts[torch.randperm(len(grid_pts), generator=g)[:10_000]].cuda()
    print(f"C1 {pts.shape[0]} grid points, whole call: %.3f ms (min %.3f)" % gpu_ms(lambda: sdf(pts), reps=30))
    for n in (1000, 100000, 2000000):
        rnd = Wk.uniform_points_device(n, bb[:, 0], bb[:, 1], 5)
        print(f"drill, {n} random points: %.3f ms (min %.3f)" % gpu_ms(lambda: sdf(rnd)))
    t0 = time.time()
    c = pv.CachedSDF("drill", 0.002, drill.bounding_box(padding=0.05), sdf, device="cuda", cache_path=None)
    torch.cuda.synchronize()
    _, gp = pv.get_coordinates_and_points_in_grid(0.002, c.ranges)
    gp = gp.cuda().float()
    print(f"cache build 0.002 m ({gp.shape[0]} voxels), mesh query only: %.3f ms (min %.3f)" % gpu_ms(lambda: sdf(gp), reps=5))
    mesh = Wk.build_c5_mesh()
    p5 = Wk.c5_points(2_000_000)
    H = torch.eye(4).unsqueeze(0).cuda()
    print("C5 chamfer 2M -> 99.5k tris: %.3f ms (min %.3f)" % gpu_ms(lambda: pv.batch_chamfer_dist(H, p5, mesh), reps=10))
    m5 = pv.MeshSDF(mesh)
    print("C5 mesh, 2M-point query with sign: %.3f ms (min %.3f)" % gpu_ms(lambda: m5(p5), reps=5))


if __name__ == "__main__":
    main()
