"""Host time per eager call: the Python fast paths against pytorch_volumetric_amd/_pvamd_fast (csrc/fastcall.cpp), same kernels."""
import ctypes, time, sys, torch
sys.path.insert(0, ".")
import workloads as Wk
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib, _pvamd_fast as F
cached = Wk.build_c2_cache()
lib = _lib.load()
fn = ctypes.cast(lib.pvamd_cached_query, ctypes.c_void_p).value
def rate(f, n=3000):
    for _ in range(300): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6
for P in (256, 15251, 1 << 20):
    pts = Wk.c2_points(cached, P, seed=3)
    val = torch.empty((P,), dtype=torch.float32, device="cuda"); grad = torch.empty((P, 3), dtype=torch.float32, device="cuda")
    desc = cached._grid_desc(); addr = ctypes.addressof(desc)
    a = rate(lambda: cached(pts)); b = rate(lambda: F.cached_call(fn, addr, 0, pts))
    c = rate(lambda: cached.query_into(pts, val, grad)); d = rate(lambda: F.cached_into(fn, addr, 0, pts, val, grad))
    v1, g1 = cached(pts); v2, g2 = F.cached_call(fn, addr, 0, pts)
    print(f"P {P}: cached(points) {a:.2f} us | fastcall {b:.2f} us || query_into {c:.2f} us | fastcall {d:.2f} us | same bits {torch.equal(v1, v2) and torch.equal(g1, g2)}")
