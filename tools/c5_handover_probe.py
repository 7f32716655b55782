"""C5 through the C-ABI with the scratch in hand: how many point groups does the chamfer kernel hand over, and what does
the call cost with / without the hand-over?"""
import os, sys, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from mesh_probe import gpu_ms
mesh = Wk.build_c5_mesh()
N = 1 << 21
pts = Wk.c5_points(N)
W = torch.eye(4).unsqueeze(0).cuda()
lib = _lib.load()
desc = mesh._mesh_desc()
sums = torch.empty((1,), dtype=torch.float64, device="cuda")
scratch = torch.zeros((_lib.mesh_scratch_bytes(N) // 8,), dtype=torch.int64, device="cuda")
order = _lib.morton_order(pts)
def run(sc):
    _lib.check(lib.pvamd_chamfer_mesh(ctypes.byref(desc), _lib.ptr(W), 1, _lib.ptr(pts), _lib.ptr(order), N, 1000.0,
                                      _lib.ptr(sums), _lib.ptr(sc), _lib.stream_ptr()), "chamfer")
t0 = gpu_ms(lambda: run(None), reps=5); ref = sums.item()
t1 = gpu_ms(lambda: run(scratch), reps=5); got = sums.item()
count = int(scratch.view(torch.int32)[0].item())
print(f"{os.environ.get('PVAMD_LIB', 'product')}: no hand-over %.3f ms | with %.3f ms | groups listed {count} of {N // 64} | rel diff {abs(got - ref) / ref:.1e}" % (t0[0], t1[0]))
