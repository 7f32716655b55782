"""The un-permute pass of the bucketed composed path on README-size link grids (C4, 200 x 262,144): the pass alone, the sorted
query kernel alone, the whole robot(points) call; and how much order the inverse permutation has (runs of consecutive
sorted positions for consecutive caller points) for random points and for a grid-ordered volume."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import workloads as Wk
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from bench_configs import gpu_time

lib = _lib.load()
robot = Wk.build_c4(0.02, 1.0)
A, P = 200, 1 << 18
robot.set_joint_configuration(Wk.c4_joint_configs(A))
comp = robot.sdf
dev = torch.device("cuda", torch.cuda.current_device())
grids = comp._leaf_grids(dev)
tfd = comp._tf_device(dev)


def runs(inv):
    d = (inv[1:].long() - inv[:-1].long()) == 1
    n_runs = int((~d).sum().item()) + 1
    return f"{d.float().mean().item() * 100:.1f} % of consecutive caller points are consecutive in the sorted order, mean run {len(inv) / n_runs:.2f}"


for name, pts in (("random", Wk.c4_points(P)),
                  ("64^3 grid order", torch.cartesian_prod(*[torch.linspace(-0.6, 0.6, 64)] * 3).cuda().contiguous())):
    P = pts.shape[0]
    _, inv, spts = _lib.morton_order(pts, min_points=0, want_inverse=True, want_sorted=True)
    Pp = -(-P // 256) * 256
    if Pp != P:
        spts = torch.cat((spts, spts[-1:].expand(Pp - P, 3))).contiguous()
    scratch = torch.empty((A, Pp, 4), dtype=torch.float32, device=dev)
    val = torch.empty((A, P), dtype=torch.float32, device=dev); grad = torch.empty((A, P, 3), dtype=torch.float32, device=dev)
    q = lambda: _lib.check(lib.pvamd_composed_query_packed(_lib.ptr(grids), 8, _lib.ptr(tfd), A, _lib.ptr(spts), Pp, _lib.ptr(scratch),
                                                           comp._query_flags, _lib.stream_ptr()), "packed")
    u = lambda: _lib.check(lib.pvamd_unpack_records(_lib.ptr(scratch), _lib.ptr(inv), P, Pp, A, _lib.ptr(val), _lib.ptr(grad),
                                                    _lib.stream_ptr()), "unpack")
    tq, _ = gpu_time(q, reps=8)
    tu, _ = gpu_time(u, reps=8)
    comp.bucket_points = True
    tw, _ = gpu_time(lambda: robot(pts), reps=8)
    comp.bucket_points = "auto"
    print(f"{os.environ.get('PVAMD_LIB', 'default')} | {name} ({P} points): sorted query {tq * 1e3:.3f} ms | un-permute {tu * 1e3:.3f} ms "
          f"({A * P * 16 / tu / 1e9:.0f} GB/s of records, {A * P / tu / 1e9:.1f} G gathers/s) | robot(points) {tw * 1e3:.3f} ms | inv: {runs(inv)}", flush=True)
