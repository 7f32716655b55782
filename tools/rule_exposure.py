"""How many results would change if the reference's third-party value-range view (multidim_indexing, un-vendored and
un-pinned: sdf.py:521,537-540) made one of the OTHER choices pvamd_grid_t.rule can express -- on the benchmark workloads:
C2 (CachedSDF 0.01 m on the drill, 1,048,576 uniform points, 52 % out of range) and C4 (RobotSDF, 200 configurations x
262,144 points).  For each alternative: the number of results that differ from the default rule's, and the largest
difference."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
import workloads as Wk

RULES = [("validity on the rounded index (range grows by half a voxel per side)", pv.RULE_VALID_ON_INDEX),
         ("round half away from zero", pv.RULE_ROUND_HALF_AWAY),
         ("floor(q + 0.5)", pv.RULE_ROUND_FLOOR_HALF),
         ("float32 range: resolution evaluated in float64", pv.RULE_RES_F64),
         ("validity on the index + half away", pv.RULE_VALID_ON_INDEX | pv.RULE_ROUND_HALF_AWAY)]


def c2(rule):
    pv.voxel.INDEX_RULE = rule
    cached = Wk.build_c2_cache()
    pts = Wk.c2_points(cached, 1 << 20, seed=1234)
    return cached(pts)[0]


def c2_f32_range(rule):  # the same cache built from python-float ranges (float32 index arithmetic): where RES_F64 applies
    pv.voxel.INDEX_RULE = rule
    obj = Wk.build_drill()
    rng = [(float(a), float(b)) for a, b in obj.bounding_box(padding=0.1)]
    cached = pv.CachedSDF("drill32", 0.01, rng, pv.MeshSDF(obj), device="cuda", cache_path=None)
    pts = Wk.c2_points(cached, 1 << 20, seed=1234)
    return cached(pts)[0]


def c4(rule):
    pv.voxel.INDEX_RULE = rule
    robot = Wk.build_c4(0.02, 0.1)
    robot.set_joint_configuration(Wk.c4_joint_configs(200))
    return robot(Wk.c4_points(1 << 18))[0]


def report(name, fn):
    base = fn(0)
    print(f"{name}: {base.numel():,} results under the default rule (round half to even, validity on the value)")
    for label, rule in RULES:
        other = fn(rule)
        differ = ~((other == base) | (other.isnan() & base.isnan()))
        n = int(differ.sum())
        worst = float((other - base).abs()[differ].max()) if n else 0.0
        print(f"   {label}: {n:,} differ ({100.0 * n / base.numel():.4f} %), largest |difference| {worst:.4g} m")
    pv.voxel.INDEX_RULE = 0


report("C2, float64 range (README flow)", c2)
report("C2, the same cache from python-float ranges (float32 index arithmetic)", c2_f32_range)
report("C4 (RobotSDF 200 x 262,144; link grids from numpy ranges)", c4)
