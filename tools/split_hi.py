import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from tests import helpers as H
drill = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
gt = pv.MeshSDF(drill)
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for n in (60_000, 100_000, 150_000, 200_000, 262_144):
    q = H.uniform_points(n, [-0.2] * 3, [0.3] * 3, seed=n).cuda()
    drill.tile_split = True; a = timed(lambda: gt(q))
    drill.tile_split = False; b = timed(lambda: gt(q))
    print(f"{n}: split {a:.3f} ms   single {b:.3f} ms")
