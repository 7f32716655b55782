"""Round 6: C4 (and variations) through the chunk-grouped composed kernel against the ungrouped one; same harness for both."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk
from bench_configs import gpu_time


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
    return min(ts)


def main():
    robot = Wk.build_c4(0.02, 0.1)
    for A, P in ((200, 1 << 18), (200, (1 << 18) + 1001), (20, 1 << 20), (2000, 1 << 15)):
        robot.set_joint_configuration(Wk.c4_joint_configs(A))
        pts = Wk.c4_points(P)
        val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
        out = {}
        for mode in (False, True):
            robot.sdf.group_points = mode
            out[mode] = graph_time(lambda: robot.sdf.query_into(pts, val, grad))
            ref = (val.clone(), grad.clone()) if mode is False else ref
        same = torch.equal(ref[0].view(torch.int32), val.view(torch.int32)) and torch.equal(ref[1].view(torch.int32), grad.view(torch.int32))
        scratch = _lib.group_points(pts)
        t_sort = graph_time(lambda: _lib.load().pvamd_group_points(_lib.ptr(pts), P, _lib.ptr(scratch), _lib.stream_ptr()))
        print(f"A {A:5d} P {P:8d}: ungrouped {out[False]:.4f} ms | grouped {out[True]:.4f} ms (sort pass {t_sort:.4f}) | same bits {same}", flush=True)
        del val, grad


if __name__ == "__main__":
    main()
