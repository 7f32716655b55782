"""How often does the composed kernel's two-minima loop enter the exact-root band (csrc/composed.hip, kNearTie)?
tools/build_variant.sh band pytorch_volumetric_amd/csrc/composed.hip -DPVAMD_COMPOSED_STATS
PVAMD_LIB=tools/variants/libpvamd_band.so python tools/band_rate.py"""
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
lib = _lib.load()


def stats(reset=True):
    buf = (ctypes.c_ulonglong * 4)()
    torch.cuda.synchronize()
    lib.pvamd_debug_band_stats(buf, 1 if reset else 0)
    return list(buf)


def report(name, fn, lanes):
    fn(); stats(); fn()
    v, bv, bl, back = stats()
    print(f"{name}: {v} 64-point leaf visits, {bv} enter the band ({100.0 * bv / max(v, 1):.3f} %), {bl} lanes in it "
          f"({100.0 * bl / max(lanes, 1):.4f} % of the (pair, leaf) visits), {back} of them keep the incumbent after the exact roots", flush=True)


robot = Wk.build_c4(0.02, 0.1)
A, P = 200, 1 << 18
robot.set_joint_configuration(Wk.c4_joint_configs(A))
pts = Wk.c4_points(P)
val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
report("C4 200 x 262,144 random points, 8 links", lambda: robot.query_into(pts, val, grad), A * P * 8)
