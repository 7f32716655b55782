"""A/B builds of the composed kernels INSIDE ONE PROCESS (same clocks, same thermal state): every library named on the command line is
loaded with ctypes next to the product one and timed in turns on the same buffers.
    python tools/ab_composed.py tools/variants/libpvamd_X.so [...]        C3 (pvamd_composed_query, 8 drills, 4M points) and
                                                                           C4 (pre-pass + grouped, 200 x 262,144)
Each figure: median over ROUNDS of (HIP events around a hipGraph of 20 calls) / 20."""
import ctypes, os, statistics, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
import workloads as Wk

ROUNDS = int(os.environ.get("AB_ROUNDS", "9"))
libs = [("product", _lib.load())]
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("pvamd_composed_query", "pvamd_composed_query_grouped", "pvamd_group_points"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    libs.append((os.path.basename(path).replace("libpvamd_", "").replace(".so", ""), lib))


def graph_of(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    return g, reps


def replay_ms(g, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def compare(title, make_call):
    graphs = [(name, graph_of(make_call(lib))) for name, lib in libs]
    for _, (g, r) in graphs:
        replay_ms(g, r)
    times = {name: [] for name, _ in graphs}
    for _ in range(ROUNDS):
        for name, (g, r) in graphs:
            times[name].append(replay_ms(g, r))
    base = statistics.median(times["product"])
    print(title + ": " + " | ".join(f"{n} {statistics.median(t):.4f} ({statistics.median(t) / base - 1:+.1%})" for n, t in times.items()), flush=True)


which = os.environ.get("AB_CASES", "c3,c4").split(",")
if "c3" in which:
    cached = Wk.build_c2_cache()
    comp = Wk.build_c3(cached)
    P = 1 << 22
    pts = Wk.c3_points(P)
    val = torch.empty((1, P), device="cuda"); grad = torch.empty((1, P, 3), device="cuda")
    grids = comp._leaf_grids(pts.device); tfd = comp._tf_device(pts.device)
    S = len(comp.sdfs)
    flags = comp._direct_flags()
    compare("C3 4M", lambda lib: (lambda: lib.pvamd_composed_query(_lib.ptr(grids), S, _lib.ptr(tfd), 1, _lib.ptr(pts), P, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr())))
    if "c3ordered" in which:
        n = 2048
        xs = torch.linspace(-0.4, 0.4, n)
        sl = torch.stack(torch.meshgrid(xs, xs, indexing="ij"), dim=-1).reshape(-1, 2)
        spts = torch.cat((sl, torch.full((n * n, 1), 0.05)), dim=1).cuda().contiguous()
        compare("C3 2048^2 ordered slice", lambda lib: (lambda: lib.pvamd_composed_query(_lib.ptr(grids), S, _lib.ptr(tfd), 1, _lib.ptr(spts), P, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr())))
    del val, grad
if "c4" in which:
    robot = Wk.build_c4(0.02, 0.1)
    A, P = 200, 1 << 18
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    val = torch.empty((A, P), device="cuda"); grad = torch.empty((A, P, 3), device="cuda")
    grids = robot.sdf._leaf_grids(pts.device); tfd = robot.sdf._tf_device(pts.device)
    scratch = _lib.group_points(pts)

    def c4_call(lib):
        def call():
            lib.pvamd_group_points(_lib.ptr(pts), P, _lib.ptr(scratch), _lib.stream_ptr())
            lib.pvamd_composed_query_grouped(_lib.ptr(grids), 8, _lib.ptr(tfd), A, _lib.ptr(scratch), P, _lib.ptr(val), _lib.ptr(grad), None, 0, _lib.stream_ptr())
        return call
    compare("C4 200 x 262144", c4_call)
if "rg" in which:  # README-size link grids (8 x 21 MB): Hilbert-sorted random points, and the README's own slice
    import numpy as np
    robot = Wk.build_c4(0.02, 1.0)
    A, P = 200, 1 << 18
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    comp = robot.sdf
    pts = Wk.c4_points(P)
    prep = comp.prepare_points(pts)
    spts = prep.sorted_points.contiguous()
    grids = comp._leaf_grids(pts.device); tfd = comp._tf_device(pts.device)
    flags = comp._query_flags | _lib.COMPOSED_NO_GROUPING
    val = torch.empty((A, P), device="cuda"); grad = torch.empty((A, P, 3), device="cuda")
    compare(f"README grids, 200 x 262144 sorted (flags {flags})", lambda lib: (lambda: lib.pvamd_composed_query(_lib.ptr(grids), 8, _lib.ptr(tfd), A, _lib.ptr(spts), P, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr())))
    compare("README grids, 200 x 262144 random", lambda lib: (lambda: lib.pvamd_composed_query(_lib.ptr(grids), 8, _lib.ptr(tfd), A, _lib.ptr(pts), P, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr())))
    _, sl = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
    sl = sl.cuda().contiguous()
    Ps = sl.shape[0]
    for A2 in (20, 200):
        robot.set_joint_configuration(Wk.c4_joint_configs(A2))
        tfd2 = comp._tf_device(sl.device)
        compare(f"README slice {A2} x {Ps}", lambda lib: (lambda: lib.pvamd_composed_query(_lib.ptr(grids), 8, _lib.ptr(tfd2), A2, _lib.ptr(sl), Ps, _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr())))
