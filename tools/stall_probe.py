"""Where do the sporadic 20-40 ms stalls of a long un-synchronised call loop come from?  (Round 2's tools/latency.py read
192 us per ComposedSDF call at P = 15,251 because ONE such stall fell into its 200-call loop: profiles/r02_probes.txt.)
Per-call host times of 30,000 calls; every call above 1 ms is listed with the CPython garbage-collector passes that
overlapped it; then the same loop with the collector frozen (gc.freeze() + gc.disable())."""
import gc, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_volumetric_amd as pv
from tests import helpers as H

obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("d", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
comp = pv.ComposedSDF([cached] * 8, pv.Transform3d(matrix=H.random_rigid(8, seed=0)))
pts = H.uniform_points(15251, [-0.2] * 3, [0.3] * 3, seed=1).cuda()

events = []
def on_gc(phase, info):
    events.append((time.perf_counter(), phase, info["generation"]))
gc.callbacks.append(on_gc)


def loop(fn, n, label):
    events.clear()
    for _ in range(100): fn()
    torch.cuda.synchronize()
    t = np.empty(n + 1)
    t[0] = time.perf_counter()
    for i in range(n):
        fn()
        t[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    d = np.diff(t) * 1e6
    slow = np.nonzero(d > 1000)[0]
    print(f"{label}: {n} calls, mean {d.mean():.1f} us, median {np.median(d):.1f} us, {len(slow)} calls above 1 ms "
          f"(sum {d[slow].sum() / 1e3:.1f} ms = {d[slow].sum() / d.sum() * 100:.0f} % of the loop)")
    for i in slow[:12]:
        inside = [(ph, g) for (ts, ph, g) in events if t[i] <= ts <= t[i + 1]]
        print(f"   call {i}: {d[i] / 1e3:.2f} ms; garbage-collector events inside it: {inside}")
    gens = [g for (_, ph, g) in events if ph == 'start']
    print(f"   collector passes during the loop: gen0 {gens.count(0)}, gen1 {gens.count(1)}, gen2 {gens.count(2)}")


loop(lambda: comp(pts), 30000, "ComposedSDF(8)(pts), P = 15,251")
loop(lambda: cached(pts), 30000, "CachedSDF(pts)")
gc.collect(); gc.freeze(); gc.disable()
loop(lambda: comp(pts), 30000, "ComposedSDF(8)(pts) with gc.freeze() + gc.disable()")
gc.enable()
