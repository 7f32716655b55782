"""Round 6 feasibility probe: what would the composed kernels gain if the points of every CHUNK of consecutive caller-order
points were regrouped spatially inside the chunk (so that a wave's 64 lanes hold neighbouring points) -- an order a workgroup
could undo through LDS, unlike the global sort of the bucketed path?  Emulated on the host side: the points are permuted
inside chunks with torch and the EXISTING kernels run on them (outputs stay in the permuted order: timing only)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from bench_configs import gpu_time


def chunk_local_sort(pts, chunk, bits):
    """Morton order on `bits` bits per axis (over the whole set's box) inside consecutive chunks of `chunk` points."""
    P = pts.shape[0]
    lo, hi = pts.min(0).values, pts.max(0).values
    q = ((pts - lo) / (hi - lo) * (1 << bits)).long().clamp_(0, (1 << bits) - 1)
    key = torch.zeros(P, dtype=torch.long, device=pts.device)
    for b in range(bits):
        for d in range(3):
            key |= ((q[:, d] >> b) & 1) << (3 * b + d)
    cid = torch.arange(P, device=pts.device) // chunk
    order = torch.argsort(cid * (1 << (3 * bits)) + key, stable=True)
    return pts[order].contiguous()


def main():
    # C4
    robot = Wk.build_c4(0.02, 0.1)
    A, P = 200, 1 << 18
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
    t_r, _ = gpu_time(lambda: robot.query_into(pts, val, grad), reps=10)
    print(f"C4 random {t_r*1e3:.4f} ms")
    for chunk in (1024, 2048, 4096, 8192, 16384, P):
        for bits in (3, 4):
            sp = chunk_local_sort(pts, chunk, bits)
            t_s, _ = gpu_time(lambda: robot.query_into(sp, val, grad), reps=10)
            print(f"C4 chunk {chunk:7d} bits {bits}: {t_s*1e3:.4f} ms")
    del val, grad
    # C3
    cached = Wk.build_c2_cache()
    comp = Wk.build_c3(cached)
    P3 = 1 << 22
    pts3 = Wk.c3_points(P3)
    v3 = torch.empty((P3,), dtype=torch.float32, device="cuda"); g3 = torch.empty((P3, 3), dtype=torch.float32, device="cuda")
    t_r, _ = gpu_time(lambda: comp.query_into(pts3, v3, g3), reps=10)
    print(f"C3 random {t_r*1e3:.4f} ms")
    for chunk in (1024, 4096, 16384, P3):
        for bits in (3, 4):
            sp = chunk_local_sort(pts3, chunk, bits)
            t_s, _ = gpu_time(lambda: comp.query_into(sp, v3, g3), reps=10)
            print(f"C3 chunk {chunk:7d} bits {bits}: {t_s*1e3:.4f} ms")


if __name__ == "__main__":
    main()
