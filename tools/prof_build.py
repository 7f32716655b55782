"""cProfile of one CachedSDF construction (C2's cache: drill, 0.01 m, padding 0.1) after a warm-up construction."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk
obj = Wk.build_drill()
def build():
    c = pv.CachedSDF("drill", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
    torch.cuda.synchronize()
    return c
build()
t0 = time.perf_counter(); build(); t1 = time.perf_counter()
print(f"second construction: {(t1 - t0) * 1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable(); build(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4200])
t0 = time.perf_counter(); o2 = Wk.build_drill(); o2.precompute_sdf() if hasattr(o2, "precompute_sdf") else None; torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"MeshObjectFactory construction: {(t1 - t0) * 1e3:.2f} ms")
