"""Name the 20-40 ms host stall of profiles/r03_stall_probe.txt: the loops of tools/latency.py from process start, every call
split into its host steps (output allocation | C-ABI launch), each step timed; every step above 2 ms is printed with what it
was.  Run it plain, and under `rocprofv3 --hip-trace --hsa-trace` (tools/r4_batch1.sh) for the API call behind the step."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from tests import helpers as H

t_start = time.perf_counter()
if os.environ.get("PVAMD_STALL_WARMUP"):
    pv.warm_up()
    print(f"pv.warm_up(): {(time.perf_counter() - t_start) * 1e3:.1f} ms", flush=True)
obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
cached = pv.CachedSDF("d", 0.01, obj.bounding_box(padding=0.1), pv.MeshSDF(obj), device="cuda", cache_path=None)
comp = pv.ComposedSDF([cached] * 8, pv.Transform3d(matrix=H.random_rigid(8, seed=0)))
mesh = pv.MeshSDF(obj)
import gc
_gc_t = [0.0]
def _on_gc(phase, info):
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    else:
        d = time.perf_counter() - _gc_t[0]
        if d > 1e-3:
            print(f"  garbage collection, generation {info['generation']}: {d * 1e3:.2f} ms, {info.get('collected')} collected, "
                  f"objects tracked {len(gc.get_objects())}, at split call {launches[0] + 1}", flush=True)
gc.callbacks.append(_on_gc)
if os.environ.get("PVAMD_STALL_NOGC"):
    gc.disable()
if os.environ.get("PVAMD_STALL_FREEZE"):
    gc.collect(); gc.freeze()
slow = []
launches = [0]
LIB = _lib.load()


def split_call(kind, pts):
    """what comp(pts) / cached(pts) do on their fast paths, step by step"""
    P = pts.shape[0]
    t0 = time.perf_counter()
    val = torch.empty((P,), dtype=torch.float32, device="cuda")
    grad = torch.empty((P, 3), dtype=torch.float32, device="cuda")
    t1 = time.perf_counter()
    if kind == "comp":
        # ComposedSDF.query_into by hand: the Python prelude and the raw C-ABI call timed apart
        dev = pts.device
        grids = comp._leaf_grids(dev)
        tfd = comp._tf_device(dev)
        args = (_lib.ptr(grids), 8, _lib.ptr(tfd), 1, _lib.ptr(pts), P, _lib.ptr(val), _lib.ptr(grad), None, comp._query_flags,
                _lib.stream_ptr())
        ta = time.perf_counter()
        rc = LIB.pvamd_composed_query(*args)
        tb = time.perf_counter()
        if tb - ta > 2e-3 or ta - t1 > 2e-3:
            print(f"  comp call {launches[0] + 1}: python prelude {(ta - t1) * 1e3:.2f} ms, raw pvamd_composed_query {(tb - ta) * 1e3:.2f} ms", flush=True)
    else:
        cached.query_into(pts, val, grad)
    t2 = time.perf_counter()
    launches[0] += 1
    if t1 - t0 > 2e-3:
        slow.append((launches[0], kind, P, "torch.empty x2", (t1 - t0) * 1e3, t0 - t_start))
    if t2 - t1 > 2e-3:
        slow.append((launches[0], kind, P, "C-ABI launch (query_into)", (t2 - t1) * 1e3, t1 - t_start))


def loop(label, fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    worst = 0.0
    for _ in range(n):
        a = time.perf_counter(); fn(); worst = max(worst, time.perf_counter() - a)
    torch.cuda.synchronize()
    print(f"  {label}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call, slowest single call {worst * 1e3:.2f} ms", flush=True)


for rnd in range(3):
    for P in (1000, 15251, 100000):
        pts = H.uniform_points(P, [-0.2] * 3, [0.3] * 3, seed=1).cuda()
        print(f"round {rnd} P={P}", flush=True)
        loop("cached split", lambda: split_call("cached", pts))
        loop("comp split", lambda: split_call("comp", pts))
        loop("cached(pts)", lambda: cached(pts))
        loop("comp(pts)", lambda: comp(pts))
        loop("mesh(pts)", lambda: mesh(pts), 20)
        loop("outside_surface", lambda: cached.outside_surface(pts))
print("steps above 2 ms:", slow if slow else "none")
