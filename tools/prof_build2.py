"""cProfile (microsecond print-out) of one CachedSDF construction (C2's cache) after warm-up, and the same for wrench@0.001."""
import cProfile, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk
obj = Wk.build_drill()
gt = pv.MeshSDF(obj)
def build():
    c = pv.CachedSDF("drill", 0.01, obj.bounding_box(padding=0.1), gt, device="cuda", cache_path=None)
    return c
for _ in range(3):
    build(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); build(); pr.disable(); torch.cuda.synchronize()
rows = sorted(pr.getstats(), key=lambda e: -e.totaltime)[:40]
for e in rows:
    code = e.code
    name = code if isinstance(code, str) else f"{os.path.basename(code.co_filename)}:{code.co_firstlineno}({code.co_name})"
    print(f"{e.totaltime*1e6:9.1f} us total {e.inlinetime*1e6:9.1f} us own  x{e.callcount:3d}  {name}")
