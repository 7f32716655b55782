import sys, os, ctypes, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_volumetric_amd as pv
from tests import helpers as H
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("d", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
PL = 1 << 26
lo = torch.tensor([r[0] for r in cached.ranges], dtype=torch.float32, device="cuda") - 0.05
hi = torch.tensor([r[1] for r in cached.ranges], dtype=torch.float32, device="cuda") + 0.05
def timeit(pts, tag, n=15):
    v = torch.empty((pts.shape[0],), device="cuda"); g = torch.empty((pts.shape[0], 3), device="cuda")
    for _ in range(3): cached.query_into(pts, v, g)
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); cached.query_into(pts, v, g); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    print(f"{tag}: median {np.median(ts):.3f} ms min {min(ts):.3f}")
big = (torch.rand((PL, 3), device="cuda") * (hi - lo) + lo).contiguous()
timeit(big, "torch.rand points (bench)")
host = (np.random.default_rng(0).random((PL, 3), dtype=np.float32) * (hi - lo).cpu().numpy() + lo.cpu().numpy()).astype(np.float32)
timeit(torch.from_numpy(host).cuda(), "numpy points uploaded")
print("bb", cached.bb, "ranges", cached.ranges)
vals = cached._packed[:, 0]
print("grid val stats", vals.min().item(), vals.max().item(), torch.isnan(cached._packed).sum().item())
# tight loop, no syncs, many reps: look for clock ramp
v = torch.empty((PL,), device="cuda"); g = torch.empty((PL, 3), device="cuda")
n = 200
starts = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; ends = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
for i in range(n):
    starts[i].record(); cached.query_into(big, v, g); ends[i].record()
torch.cuda.synchronize()
ts = np.array([s.elapsed_time(e) for s, e in zip(starts, ends)])
print("tight loop: first10", ts[:10].round(3), "last10", ts[-10:].round(3), "median", np.median(ts))
# interleave with a big torch copy kernel like kbench's round robin
for i in range(20):
    tmp = g.clone()
    starts[i].record(); cached.query_into(big, v, g); ends[i].record()
torch.cuda.synchronize()
print("interleaved with clone:", np.median([s.elapsed_time(e) for s, e in zip(starts[:20], ends[:20])]))
