"""Composed kernel variants side by side in one process (round 4): dispatcher's choice (0), wave-tile with the two-minima leaf
loop (4), wave-tile with the round-3 leaf loop (4 | 16), one point per lane (2) -- C3, C4 (100 KB grids), the README case and
README-size grids, random and Morton-sorted points.  ms per call: the best of three interleaved rounds of a 12-call median (HIP events)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
import torch
import workloads as Wk
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from bench_configs import gpu_time

VARIANTS = ((0, "auto"), (4, "wave-tile split-minima"), (4 | 16, "wave-tile round-3 loop"), (2, "per-lane split-minima"), (2 | 16, "per-lane round-3 loop"))


def run(name, sdf, pts, A, variants=VARIANTS):
    P = pts.shape[0]
    val = torch.empty((A, P), dtype=torch.float32, device="cuda"); grad = torch.empty((A, P, 3), dtype=torch.float32, device="cuda")
    comp = getattr(sdf, "sdf", sdf)
    comp._leaf_grids(pts.device)
    base = comp._query_flags
    best = {}
    for rnd in range(3):  # interleaved rounds: the first thing measured after a pause reads ~10 % slow (clocks)
        for fl, label in variants:
            comp._query_flags = base | fl
            t, tmin = gpu_time(lambda: sdf.query_into(pts, val, grad), warm=4, reps=12)
            best[label] = min(best.get(label, 1e9), t)
    comp._query_flags = base
    print(f"{name} [base flags {base}]: " + " | ".join(f"{label} {best[label] * 1e3:.4f}" for _, label in variants), flush=True)


which = sys.argv[1:] or ["c3", "c4", "readme", "big"]
if "c3" in which:
    cached = Wk.build_c2_cache()
    comp = Wk.build_c3(cached)
    p3 = Wk.c3_points(1 << 22)
    run("C3 4M random", comp, p3, 1)
    run("C3 4M sorted", comp, p3[_lib.morton_order(p3).long()].contiguous(), 1)
if "c4" in which:
    robot = Wk.build_c4(0.02, 0.1)
    robot.set_joint_configuration(Wk.c4_joint_configs(200))
    p4 = Wk.c4_points(1 << 18)
    run("C4 200x262144 pad 0.1 random", robot, p4, 200)
    run("C4 200x262144 pad 0.1 sorted", robot, p4[_lib.morton_order(p4).long()].contiguous(), 200)
    _, slice_pts = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
    slice_pts = slice_pts.cuda()
    run("README slice 200x15251 pad 0.1", robot, slice_pts, 200)
    robot.set_joint_configuration(Wk.c4_joint_configs(20))
    run("README slice 20x15251 pad 0.1", robot, slice_pts, 20)
if "big" in which:
    robot = Wk.build_c4(0.02, 1.0)
    robot.set_joint_configuration(Wk.c4_joint_configs(200))
    p4 = Wk.c4_points(1 << 18)
    # base flags = inline exact (1): the queued loop is not used there; 4 and 4 | 16 are the same kernel
    run("C4 200x262144 pad 1.0 random", robot, p4, 200, ((0, "auto"), (2, "per-lane")))
    run("C4 200x262144 pad 1.0 sorted", robot, p4[_lib.morton_order(p4).long()].contiguous(), 200, ((0, "auto"), (2, "per-lane")))
    _, slice_pts = pv.get_coordinates_and_points_in_grid(0.01, np.array([[-1, 0.5], [0.02, 0.02], [-0.2, 0.8]]))
    slice_pts = slice_pts.cuda()
    per_lane = ((0, "auto"), (2, "per-lane split-minima"), (2 | 16, "per-lane round-3 loop"))
    run("README slice 200x15251 pad 1.0", robot, slice_pts, 200, per_lane)
    rp = Wk.c4_points(15251)
    run("README-size random 200x15251 pad 1.0", robot, rp, 200, per_lane)
    robot.set_joint_configuration(Wk.c4_joint_configs(20))
    run("README slice 20x15251 pad 1.0", robot, slice_pts, 20, per_lane)
