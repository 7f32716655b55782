"""The legs of bench.py -- BASELINE.json's other configs (C1, C3, C4, C5), the cache builds, the reference README's case -- and
the timing / roofline helpers they share with the headline.  Imported by bench.py; nothing here touches oracle/."""
import json
import os
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X datasheet (MI355X_MICROARCH.md); 6.3-6.5 TB/s is what a float4 copy reaches (profiles/)
BYTES_PER_QUERY = 28   # 12 B point read + 4 B value + 12 B gradient written
BYTES_PER_PAIR_C4 = 16  # C4: 4 B value + 12 B gradient written per (configuration, point); points are re-read from L2
XGMI_LINK_GBS = 153.0   # per direction per peer link (7 links per GPU on an 8-GPU node; the task's hardware notes)
N_SIMD = 1024
FP32_PEAK_TFLOPS = 157.3   # MI355X vector fp32
FLOP_PER_EXACT_PAIR = 80.0  # one point-triangle closest-point (or ray) test, nominal
README_PUBLISHED_MS = {20: 37.688577, 200: 128.645445}  # /root/reference README.md:196-200, RTX 2080 Ti, KUKA iiwa (8 links)


def capture_graph(torch, fn, n):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    return graph


def graph_ms_per_launch(torch, graph, n, reps=3, stats=False):
    """HIP events on the launch stream around one replay of an n-launch graph, / n: MEAN over `reps` replays (and, with
    stats, also the best one -- round 3 printed the best under the name "mean")."""
    each = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        each.append(e0.elapsed_time(e1) / n)
    mean = sum(each) / len(each)
    return (mean, min(each)) if stats else mean


def time_eager_kernel(torch, np, fn, reps):
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    torch.cuda.synchronize()
    for i in range(reps):
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    return float(np.mean(ms)), float(np.median(ms)), float(np.min(ms))


def time_calls(torch, np, fn, reps=400):
    """The drop-in call as a user makes it: (a) `reps` calls back to back, one synchronize at the end (what a loop that
    consumes the results on the GPU sees), (b) synchronize after every call (what a loop that reads them back sees)."""
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    back_to_back = (time.perf_counter() - t0) / reps
    each = []
    for _ in range(min(reps, 200)):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        each.append(time.perf_counter() - t0)
    return back_to_back * 1e3, float(np.median(each)) * 1e3


PROFILE_ROUNDS = ("r06", "r05", "r04")  # committed profile files are looked up newest round first


def profile_path(suffix):
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{suffix}")
        if os.path.exists(path):
            return path
    return None


def call_latency(torch, np, fn, calls=20_000, drain_every=256):
    """Host time of every single call of a loop (perf_counter around the call; the launch is asynchronous), the queue drained
    every `drain_every` calls outside the timed span: what a planner that issues one query per step sees, incl. the tail."""
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t = np.empty(calls)
    pc = time.perf_counter
    for i in range(calls):
        a = pc()
        fn()
        t[i] = pc() - a
        if i % drain_every == drain_every - 1:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    us = t * 1e6
    return {"calls": calls, "p50_us": float(np.percentile(us, 50)), "p99_us": float(np.percentile(us, 99)),
            "p99_9_us": float(np.percentile(us, 99.9)), "max_us": float(us.max()), "calls_above_1ms": int((us > 1000).sum())}


def valu_session():
    path = profile_path("valu_session.json")
    return (json.load(open(path)), os.path.basename(path)) if path else (None, None)


def valu_model():
    """Once per record: how every VALU roofline below is formed (the per-leg objects carry numbers only)."""
    prof, name = valu_session()
    return {"file": None if name is None else f"profiles/{name}", "n_simd": N_SIMD,
            "issue_rates_ns": None if prof is None else {k: v for k, v in prof["issue_rates_ns"].items()
                                                         if isinstance(v, (int, float))},
            "ceiling": "1024 SIMDs / max(f_slow * t_slow, t_best) wave64 VALU inst/s, with the call's own dynamic opcode mix",
            "f_slow": "share of VALU instructions that are not an f32 add / mul / fma (SQ_INSTS_VALU_* class counters)",
            "frac": "SQ_INSTS_VALU and kernel time of ONE rocprofv3 session (tools/valu_session.sh) against that ceiling",
            "frac_this_run": "the same instruction count over THIS run's time"}


def valu_roofline(kernel_key, ms, launches_per_step=1):
    """Roofline object of a vector-ALU-bound leg.  Every number behind `frac` comes from ONE committed rocprofv3 session
    (profiles/rNN_valu_session.json): SQ_INSTS_VALU per call, the kernels' time in that session, the dynamic opcode mix and the
    issue rates tools/valu_rate.bin measured minutes apart on the same box.  `frac_this_run` divides the session's instruction
    count by this run's time instead (numerator and time from different sessions: secondary)."""
    prof, name = valu_session()
    try:
        entry = prof["workloads"][kernel_key]
        rates = prof["issue_rates_ns"]
    except Exception as exc:
        return {"bound": "valu", "achieved": None, "frac": None, "note": f"no committed session for {kernel_key}: {exc!r}"[:120]}
    inst = entry["SQ_INSTS_VALU"] * launches_per_step
    f_slow = entry["mix"]["slow_fraction"]
    # ns per wave64 instruction per SIMD at which THIS mix can issue at best: its slow-group share alone, or the fastest stream
    # the micro-benchmark found at all (fast and slow opcodes issue side by side: a weighted sum of the two is NOT a ceiling)
    t_mix = max(f_slow * rates["slow"], rates["best_any"])
    peak = N_SIMD / (t_mix * 1e-9)
    sess_ms = entry["kernel_ms_same_session"] * launches_per_step
    achieved = inst / (sess_ms * 1e-3)
    return {"bound": "valu", "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "G wave64 VALU inst/s",
            "frac": achieved / peak, "frac_this_run": inst / (ms * 1e-3) / peak, "util_valu": inst / (ms * 1e-3) / peak,
            "util_valu_is": "issued VALU instructions (committed session) / this run's time / own-mix issue ceiling: a utilisation",
            "kernel_ms_same_session": sess_ms,
            "valu_inst_per_step": inst, "active_lanes_per_inst": entry.get("active_lanes"), "slow_fraction": f_slow,
            "session": f"profiles/{name}[workloads][{kernel_key}]"}



def exact_pairs_of(torch, obj, fn):
    """Exact point-triangle tests (closest-point + ray) ONE untimed run of `fn` executes over mesh object `obj`: counted live by
    the product kernels (pvamd_mesh_t.pair_counters), detached again before anything is timed."""
    counters = obj.count_exact_pairs(True)
    try:
        fn()
        torch.cuda.synchronize()
        closest, rays = (int(x) for x in counters.tolist())
    finally:
        obj.count_exact_pairs(False)
    return closest, rays


def frac_8d_pairs(pairs, ms):
    """SURVEY.md 8(d), mesh configs: exact tests executed x 80 flop / time / 157.3 TFLOP/s (vector fp32 peak)."""
    return pairs * FLOP_PER_EXACT_PAIR / (ms * 1e-3) / (FP32_PEAK_TFLOPS * 1e12)


LEG_REGIONS = 3  # timed regions per leg figure; the median one is reported (one host stall inside a single region read as 16 ms / step once)


def median_region(timer, fn, regions=LEG_REGIONS):
    """`regions` timed regions of `fn` (each: barrier + synchronize on both sides, MAX over ranks), the median one.  Every rank runs
    the same number of regions: the timer holds collectives."""
    times = sorted(timer(fn) for _ in range(regions))
    return times[len(times) // 2]


class LegSkipped(Exception):
    pass


def settle(torch, fn, seconds=0.3, at_least=3):
    """Untimed warm-up of a leg: call `fn` until `seconds` of wall time have gone by (and at least `at_least` times), then
    synchronize.  A leg's three warm-up calls of round 3 ended before the chip had left its idle clocks (C4: 0.79 ms in the leg
    against 0.70 ms for the same kernel in a busy A/B loop, tools/composed_ab.py)."""
    t0, n = time.perf_counter(), 0
    while n < at_least or time.perf_counter() - t0 < seconds:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def build_robot(Wk, cache, padding):
    if padding not in cache:
        cache[padding] = Wk.build_c4(resolution=0.02, padding=padding)
    return cache[padding]


def leg_c4(torch, dist, Wk, pv, timer, gate, robots, rank, world, steps, padding, with_gather, small=False, use_pg=False):
    """BASELINE configs[3]: RobotSDF (7-DOF, 8 links), A=200 joint configurations x P=262,144 points, the POINTS
    sharded over the ranks (strong scaling: the total work is fixed).  Leg 1 leaves (val, grad) sharded -- no
    collective; leg 2 times ShardedSDF.__call__: query into packed records + ONE RCCL all-gather + the unpack kernel that
    writes (A, P[, 3]) order (model_to_sdf.py:117-125 on every rank's slice); leg 3 shards the CONFIGURATIONS instead
    (rows gathered in place, no unpack: SURVEY.md 8(e)).  100 KB link grids: the allocation-free entry (query_into, what a
    planner loop calls).  README-size grids: the drop-in call robot(points), which sorts the shared point set along a Hilbert
    curve once per call and un-permutes (ComposedSDF.bucket_points), output allocation included; for those the leg also times
    the prepared form (prepare_points once, query_prepared per step: no sort; order="sorted" also drops the un-permute)."""
    A, P = (8, 1 << 14) if small else (200, 1 << 18)
    robot = build_robot(Wk, robots, padding)
    robot.set_joint_configuration(Wk.c4_joint_configs(A))
    pts = Wk.c4_points(P)
    start, stop, chunk = pv.shard_range(P, world, rank)
    mine = pts[start:stop].contiguous()
    n = mine.shape[0]
    val = torch.empty((A, n), dtype=torch.float32, device="cuda")
    grad = torch.empty((A, n, 3), dtype=torch.float32, device="cuda")

    bucketed = robot.sdf._bucketing_pays(A, n, mine)
    if bucketed:
        def one_step():
            return robot(mine)
    else:
        def one_step():
            robot.query_into(mine, val, grad)

    def sharded_steps():
        for _ in range(steps):
            one_step()

    settle(torch, one_step)
    gate()
    t = median_region(timer, sharded_steps)
    pairs = A * P * steps
    out = {"config": f"C4: RobotSDF 8 links, grids res 0.02 padding {padding}, A={A} x P={P}, points sharded x{world}",
           "scaling": "strong", "n_gpus": world, "steps": steps, "unit": "(configuration, point) pairs/s",
           "link_grid_voxels": [int(s._packed.shape[0]) for s in robot.sdf.sdfs],
           "call": "robot(points): Hilbert sort + fused kernel + un-permute, outputs allocated" if bucketed
                   else "robot.query_into(points, val, grad): fused kernel, caller's buffers",
           "sharded": {"gather": False, "value": pairs / t, "ms_per_step": t / steps * 1e3,
                       "hbm_write_GBs": BYTES_PER_PAIR_C4 * pairs / t / 1e9,
                       "frac_of_hbm_peak": BYTES_PER_PAIR_C4 * pairs / t / 1e9 / (HBM_PEAK_GBS * world),
                       # 16 B written per (configuration, point) pair against 8 TB/s per GPU -- this run's time only
                       "frac_8d": BYTES_PER_PAIR_C4 * pairs / t / 1e9 / (HBM_PEAK_GBS * world),
                       "frac_8d_is": "16 B x A x P / ms_per_step / (8 TB/s x n_gpus)",
                       "kernel": "group_points_kernel + composed_query_grouped" if robot.sdf._grouping_pays(A, n, robot.sdf._query_flags)
                                 else "composed_query_wave / composed_query_scalar"}}
    if not bucketed and not small and world == 1:
        out["sharded"]["roofline"] = valu_roofline("c4_composed_query_wave", t / steps * 1e3)
    if bucketed:
        out["sharded"]["bound"] = "L1->L2 request rate (16-byte gathers; profiles/r04_unpermute.txt, r03_cq64_counters.md)"
        if hasattr(robot.sdf, "prepare_points"):
            handle = robot.sdf.prepare_points(mine)
            for order in ("sorted", "caller"):
                def prepared_step(order=order):
                    return robot.sdf.query_prepared(handle, order=order)

                settle(torch, prepared_step, seconds=0.1)
                tp = median_region(timer, lambda: [prepared_step() for _ in range(steps)])
                out["sharded"][f"prepared_{order}_ms"] = tp / steps * 1e3
            v0, g0 = robot(mine)
            v1, g1 = robot.sdf.query_prepared(handle, order="caller")
            v2, g2 = robot.sdf.query_prepared(handle, order="sorted")
            idx = handle.order.long()
            out["sharded"]["prepared_equals_direct"] = bool(torch.equal(v0, v1) and torch.equal(g0, g1) and
                                                            torch.equal(v0[:, idx], v2) and torch.equal(g0[:, idx], g2))
            del v0, g0, v1, g1, v2, g2, handle
    if with_gather and (world > 1 or use_pg):
        sharded = pv.ShardedSDF(robot, gather=True, compute_device=torch.device("cuda"))
        gsteps = max(2, steps // 4)
        full = None

        def gather_steps():
            nonlocal full
            for _ in range(gsteps):
                full = sharded(pts)

        sharded(pts)
        tg = median_region(timer, gather_steps)
        ref = robot(pts[:65536])  # after timing: the gathered result against the unsharded call, bit for bit
        same = bool(torch.equal(full[0][:, :65536], ref[0]) and torch.equal(full[1][:, :65536], ref[1]))
        recv = getattr(sharded, "bytes_received_per_rank", None)
        # self-check of the first multi-GPU line: as many ranks as GPUs asked for, the RCCL backend, and a rank receives
        # (W - 1) / W of the packed output (the 256-point padding of every rank's slice included)
        padded = -(-(-(-P // world)) // 256) * 256
        want_recv = (world - 1) * A * padded * 16
        check = {"ranks": dist.get_world_size(), "ranks_expected": world, "backend": dist.get_backend(),
                 "bytes_received_per_rank_expected": want_recv, "bytes_match": recv == want_recv,
                 "kernel_only_ms": t / steps * 1e3, "gathered_ms": tg / gsteps * 1e3}
        check["ok"] = bool(check["ranks"] == world and same and check["bytes_match"] and
                           (check["backend"] == "nccl" or world == 1 or getattr(sharded, "last_path", "") != "packed"))
        out["gathered"] = {"gather": True, "value": A * P * gsteps / tg, "ms_per_step": tg / gsteps * 1e3, "steps": gsteps,
                           "self_check": check,
                           "collective": f"packed (val, grad) records, all_gather_into_tensor x1 ({dist.get_backend()}), unpack kernel",
                           "bytes_received_per_rank": recv,
                           "xgmi_lower_bound_ms": None if not recv or world < 2 else recv / (world - 1) / (XGMI_LINK_GBS * 1e9) * 1e3,
                           "xgmi_model": f"full mesh: each of {world - 1} peers sends its slab over its own {XGMI_LINK_GBS:.0f} GB/s link",
                           "path": getattr(sharded, "last_path", None), "equals_unsharded_call": same,
                           "output_shape": [list(full[0].shape), list(full[1].shape)]}
        if not bucketed:
            by_cfg = pv.ShardedSDF(robot, gather=True, compute_device=torch.device("cuda"), shard="configs")
            cfull = None

            def config_steps():
                nonlocal cfull
                for _ in range(gsteps):
                    cfull = by_cfg(pts)

            by_cfg(pts)
            tc = median_region(timer, config_steps)
            same_c = bool(torch.equal(cfull[0][:, :65536], ref[0]) and torch.equal(cfull[1][:, :65536], ref[1]))
            out["gathered_by_configs"] = {"gather": True, "shard": "configs", "value": A * P * gsteps / tc,
                                          "ms_per_step": tc / gsteps * 1e3, "steps": gsteps,
                                          "collective": f"all_gather_into_tensor x2 (val rows, grad rows; {dist.get_backend()}), no unpack",
                                          "bytes_received_per_rank": getattr(by_cfg, "bytes_received_per_rank", None),
                                          "equals_unsharded_call": same_c}
    else:
        out["gathered"] = None
    return out


def leg_readme(torch, np, Wk, pv, timer, gate, robots, rank, world, A):
    """The reference README's own benchmark shape (/root/reference README.md:150-200): link caches at resolution 0.02 with
    padding 1.0, A joint configurations (q0 + 0.1 N(0,1)), the README's M = 15,251 query points -- the ordered 151 x 1 x
    101 slice `get_coordinates_and_points_in_grid(0.01, [[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]])` -- through the drop-in
    call `robot(points)` (output allocation included).  Every rank runs the whole case (replicas: it is far too small to
    shard).  NOT like for like with the published figure: synthetic 7-DOF arm (the KUKA assets are not available
    offline), one MI355X against an RTX 2080 Ti.  `configure_plus_query_ms` = robot.configure_and_query_into(q_on_gpu, points,
    val, grad) back to back (pvamd_configure_chain: sin / cos + FK + offset^-1 o world^-1 in ONE launch, then the query);
    `_graph_ms` = the same two kernels replayed from a hipGraph (HIP events / 100)."""
    robot = build_robot(Wk, robots, 1.0)
    th = Wk.c4_joint_configs(A)
    _, pts = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
    pts = pts.cuda()
    M = pts.shape[0]
    robot.set_joint_configuration(th)
    gate()
    call_ms, synced_ms = time_calls(torch, np, lambda: robot(pts), reps=200)
    sjc_ms, _ = time_calls(torch, np, lambda: robot.set_joint_configuration(th), reps=100)       # joint values on the host
    th_dev = th.cuda().contiguous()
    sjc_dev_ms, _ = time_calls(torch, np, lambda: robot.set_joint_configuration(th_dev), reps=100)  # already on the GPU
    val, grad = robot(pts)
    both_ms, _ = time_calls(torch, np, lambda: robot.configure_and_query_into(th_dev, pts, val, grad), reps=200)
    g = capture_graph(torch, lambda: robot.configure_and_query_into(th_dev, pts, val, grad), 100)
    g.replay()
    torch.cuda.synchronize()
    both_graph_ms = graph_ms_per_launch(torch, g, 100)
    del g
    return {"config": f"README case: RobotSDF 8 links, grids res 0.02 padding 1.0, A={A} x M={M} slice points, robot(points)",
            "scaling": "replicas", "n_gpus": world, "unit": "ms per robot(points) call",
            "ms_per_call": call_ms, "ms_per_call_synchronized_each": synced_ms, "pairs_per_s": A * M / (call_ms * 1e-3),
            "set_joint_configuration_ms": sjc_ms, "set_joint_configuration_device_q_ms": sjc_dev_ms,
            "configure_plus_query_ms": both_ms, "configure_plus_query_graph_ms": both_graph_ms,
            "output_shapes": [list(val.shape), list(grad.shape)],
            "published_ms": README_PUBLISHED_MS.get(A), "published_on": "RTX 2080 Ti, KUKA iiwa (README.md:196-200)",
            "like_for_like": False, "why_not": "synthetic 7-DOF arm with 8 ellipsoid links (no KUKA assets offline); different GPU"}


def leg_c3(torch, Wk, pv, timer, gate, cached, rank, world, steps, small=False):
    """BASELINE configs[2]: ComposedSDF of 8 transformed drills (the C2 cache under 8 rigid transforms), 4,194,304 query
    points, transform + lookup + min over leaves fused (sdf.py:392-433); the points sharded over the ranks, results left
    sharded.  28 B/query algorithmic, but 8 leaf visits per point bound it: the vector ALUs, not HBM."""
    P = (1 << 16) if small else (1 << 22)
    comp = Wk.build_c3(cached)
    pts = Wk.c3_points(P)
    start, stop, _ = pv.shard_range(P, world, rank)
    mine = pts[start:stop].contiguous()
    n = mine.shape[0]
    val = torch.empty((1, n), dtype=torch.float32, device="cuda")
    grad = torch.empty((1, n, 3), dtype=torch.float32, device="cuda")

    def run():
        for _ in range(steps):
            comp.query_into(mine, val, grad)

    settle(torch, run)
    gate()
    t = median_region(timer, run)
    gbs = BYTES_PER_QUERY * P * steps / t / 1e9
    return {"config": f"C3: ComposedSDF of 8 transformed drills (37x33x40 cache each), {P} points, points sharded x{world}",
            "scaling": "strong", "n_gpus": world, "steps": steps, "unit": "queries/s", "value": P * steps / t,
            "ms_per_step": t / steps * 1e3, "call": "comp.query_into(points, val, grad): fused kernel, caller's buffers",
            "frac_8d": gbs / (HBM_PEAK_GBS * world), "frac_8d_is": "28 B x P / ms_per_step / (8 TB/s x n_gpus)  (SURVEY.md 8(d))",
            "roofline": dict(valu_roofline("c3_composed_query", t / steps * 1e3) if (world == 1 and not small) else {"bound": "valu"},
                             hbm_algorithmic_GBs=gbs, frac_of_hbm_peak=gbs / (HBM_PEAK_GBS * world))}


def leg_c1(torch, np, Wk, pv, gate, world):
    """BASELINE configs[0]: MeshSDF on the YCB drill (15,728 triangles), 10,000 of the 0.002 m grid points (the reference's
    tests/test_sdf.py:46-48) -- the reference's CPU path (Embree) config, here one call of the GPU mesh query.  Every rank runs
    the whole case (replicas).  A 10,000-point call is latency- as much as throughput-bound (dependent launches): the VALU
    roofline says how busy the ALUs are, not that they are the limit."""
    drill = Wk.build_drill()
    sdf = pv.MeshSDF(drill)
    _, grid_pts = pv.get_coordinates_and_points_in_grid(0.002, drill.bounding_box(0.01))
    pts = grid_pts[torch.randperm(len(grid_pts), generator=torch.Generator().manual_seed(0))[:10_000]].cuda()
    sdf(pts)
    closest, rays = exact_pairs_of(torch, drill, lambda: sdf(pts))
    gate()
    call_ms, synced_ms = time_calls(torch, np, lambda: sdf(pts), reps=100)
    return {"config": f"C1: MeshSDF on YcbPowerDrill ({drill.num_faces} triangles), 10,000 grid points, one call",
            "scaling": "replicas", "n_gpus": world, "unit": "points/s", "value": 10_000 / (call_ms * 1e-3),
            "ms_per_step": call_ms, "ms_per_call": call_ms, "ms_per_call_synchronized_each": synced_ms,
            "exact_pairs_per_step": closest + rays, "exact_closest_tests": closest, "exact_ray_tests": rays,
            "frac_8d": frac_8d_pairs(closest + rays, call_ms),
            "frac_8d_is": "exact point-triangle tests executed (counted live) x 80 flop / ms_per_step / 157.3 TFLOP/s",
            "brute_force_equivalent_pairs_per_s": 10_000 * drill.num_faces / (call_ms * 1e-3),
            "roofline": valu_roofline("c1_mesh_query", call_ms)}


def leg_c5(torch, dist, Wk, pv, timer, gate, rank, world, steps, small=False, use_pg=False):
    """BASELINE configs[4]: unidirectional chamfer, 2,097,152 source points -> 99,500-triangle mesh, the source
    points sharded over the ranks; each rank reduces its slice, then ONE all-reduce of B float64 partial sums (+ the
    count) -- chamfer.py:79-94 with the mean taken over the global N.  `brute_force_equivalent_pairs_per_s` = pairs a plain
    double loop would evaluate: the kernel culls, so it is a throughput equivalent, not work done."""
    N = (1 << 16) if small else (1 << 21)
    mesh = Wk.build_c5_mesh()
    pts = Wk.c5_points(N)
    W = torch.eye(4).unsqueeze(0).cuda()
    err = None

    def run():
        nonlocal err
        for _ in range(steps):
            if world > 1 or use_pg:
                err = pv.sharded_chamfer(W, pts, obj_factory=mesh, scale=1000.0)
            else:
                err = pv.batch_chamfer_dist(W, pts, obj_factory=mesh, scale=1000.0)

    mesh._mesh_desc()  # upload + prepare the mesh: set-up, before the gate
    closest, rays = exact_pairs_of(torch, mesh, lambda: pv.batch_chamfer_dist(W, pts, obj_factory=mesh, scale=1000.0))
    gate()
    # (with ranks the warm-up count must be the same on every rank: `run` holds a collective)
    settle(torch, run, seconds=0.0 if (world > 1 or use_pg) else 0.3, at_least=2 if (world > 1 or use_pg) else 1)
    t = median_region(timer, run)
    F = mesh.num_faces
    analytic = float((((pts.norm(dim=-1) - 0.1) * 1000.0) ** 2).mean())
    return {"config": f"C5: chamfer, {N} points -> {F}-triangle sphere mesh, points sharded x{world}, B=1",
            "scaling": "strong", "n_gpus": world, "steps": steps, "unit": "points/s", "value": N * steps / t,
            "ms_per_step": t / steps * 1e3, "brute_force_equivalent_pairs_per_s": N * F * steps / t,
            "exact_pairs_per_step": closest + rays, "exact_closest_tests": closest, "exact_ray_tests": rays,
            "frac_8d": frac_8d_pairs(closest + rays, t / steps * 1e3) / world,
            "frac_8d_is": "exact tests executed over all N points (counted live) x 80 flop / ms_per_step / (157.3 TF x n_gpus)",
            "roofline": valu_roofline("c5_chamfer_mesh", t / steps * 1e3) if (world == 1 and not small) else None,
            "collective": None if (world == 1 and not use_pg) else f"all_reduce of B=1 float64 sums + count ({dist.get_backend()})",
            "chamfer_mm2": float(err[0]), "analytic_sphere_mm2": analytic,
            "rel_err_vs_analytic": abs(float(err[0]) - analytic) / analytic}


CACHE_BUILDS = (("drill_0.01", "ycb_power_drill.npz", 0.01, 0.1, (37, 33, 40)),        # README.md:47 (the C2 cache)
                ("drill_0.002", "ycb_power_drill.npz", 0.002, 0.01, (92, 73, 105)),    # tests/test_sdf.py:46
                ("wrench_0.001", "offset_wrench_nogrip.obj", 0.001, 0.05, (218, 126, 111)))  # tests/test_model_to_sdf.py:272


def leg_cache_build(torch, np, Wk, pv, gate, world, small=False):
    """CachedSDF construction with the cache filled on the device -- coordinates, the mesh kernel over every
    voxel centre (sdf.py:498-516 on the GPU instead of Embree on the host), packing into 16-byte records; cache_path=None, so
    no pickle round trip.  Per build: wall time of the CachedSDF(...) call (median of 3, after one untimed build that grows
    the allocator) and voxel centres per second.  Every rank builds all of them (replicas)."""
    out = {"scaling": "replicas", "n_gpus": world, "unit": "voxel centres/s", "builds": {}}
    gate()
    for key, mesh_name, res, pad, want_shape in CACHE_BUILDS[:1] if small else CACHE_BUILDS:
        obj = pv.MeshObjectFactory(Wk.mesh_path(mesh_name))
        gt = pv.MeshSDF(obj)
        gt(torch.zeros(64, 3).cuda())  # mesh upload + preparation: not part of a cache build's repeatable cost
        def build():
            return pv.CachedSDF(key, res, obj.bounding_box(padding=pad), gt, device="cuda", cache_path=None)

        closest, rays = exact_pairs_of(torch, obj, build)
        times, gpu_times = [], []
        for i in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            c = build()
            e1.record()
            torch.cuda.synchronize()
            if i:
                times.append(time.perf_counter() - t0)
                gpu_times.append(e0.elapsed_time(e1))
        shape = tuple(int(x) for x in c._view.shape)
        n = int(np.prod(shape))
        ms, gpu_ms = float(np.median(times)) * 1e3, float(np.median(gpu_times))
        entry = {"mesh": mesh_name, "triangles": obj.num_faces, "resolution": res, "padding": pad, "grid": list(shape),
                 "grid_is_the_reference_size": shape == want_shape, "voxels": n, "ms": ms, "value": n / (ms * 1e-3),
                 "ms_is": "wall time of CachedSDF(...) incl. the final synchronize, median of 5",
                 "gpu_ms": gpu_ms, "gpu_ms_is": "HIP events around the same call: first launch to last kernel done",
                 "exact_pairs": closest + rays, "exact_closest_tests": closest, "exact_ray_tests": rays,
                 "frac_8d": frac_8d_pairs(closest + rays, gpu_ms),
                 "frac_8d_is": "exact point-triangle tests executed (counted live) x 80 flop / gpu_ms / 157.3 TFLOP/s"}
        if world == 1 and not small:
            roof = valu_roofline(f"build_{key}", gpu_ms)
            if roof.get("frac") is not None:
                entry["roofline"] = roof
        out["builds"][key] = entry
        del c
        torch.cuda.empty_cache()
    return out
