"""Is C5 (2M points -> 99,500-triangle sphere) bound by its slowest point groups (those near the sphere's centre)?
Same points, same Morton grouping, but the groups nearest the centre are dispatched FIRST instead of mid-way."""
import os, sys, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from bench_configs import gpu_time
mesh = Wk.build_c5_mesh()
N = 1 << 21
pts = Wk.c5_points(N)
W = torch.eye(4).unsqueeze(0).cuda()
lib = _lib.load()
desc = mesh._mesh_desc()
sums = torch.empty((1,), dtype=torch.float64, device="cuda")
scratch = torch.empty((_lib.mesh_scratch_bytes(N) // 8,), dtype=torch.int64, device="cuda")  # None: nothing is handed over
def run(order):
    _lib.check(lib.pvamd_chamfer_mesh(ctypes.byref(desc), _lib.ptr(W), 1, _lib.ptr(pts), _lib.ptr(order), N, 1000.0,
                                      _lib.ptr(sums), _lib.ptr(scratch), _lib.stream_ptr()), "chamfer")
order = _lib.morton_order(pts)
t0, _ = gpu_time(lambda: run(order), reps=5); ref = sums.item()
# group-level reorder: groups of 64 consecutive points in Morton order, sorted by their mean distance to the centre
g = pts[order.long()].reshape(-1, 64, 3).norm(dim=-1).mean(dim=1)
gorder = torch.argsort(g)  # centre first
order2 = order.reshape(-1, 64)[gorder].reshape(-1).contiguous()
t1, _ = gpu_time(lambda: run(order2), reps=5); r1 = sums.item()
order3 = order.reshape(-1, 64)[gorder.flip(0)].reshape(-1).contiguous()  # centre last
t2, _ = gpu_time(lambda: run(order3), reps=5)
print(f"C5: Morton order {t0*1e3:.2f} ms | heaviest (centre) groups first {t1*1e3:.2f} ms | centre last {t2*1e3:.2f} ms | same sum {abs(ref-r1)/ref:.1e}")
