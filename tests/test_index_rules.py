"""The choices of the reference's value-range view that nothing pins (multidim_indexing is an un-vendored dependency; call
sites sdf.py:521,537-540) are a FIELD of the grid descriptor (pvamd_grid_t.rule), with an oracle twin each:
validity on the value | on the rounded index; round half to even | half away | floor(q + 0.5); a float32 range's resolution
in float32 | float64.  CPU: the oracle's statements on hand-checked numbers + pvamd_grid_finalize's valid interval;
GPU (-m gpu): every kernel that looks a voxel up, against the oracle, for each rule, on points sprayed at the half-voxel
planes and range edges where the rules differ."""
import ctypes
import itertools

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from pytorch_volumetric_amd import _lib, voxel
from tests import helpers as H

ON_INDEX, HALF_AWAY, FLOOR_HALF, RES_F64 = pv.RULE_VALID_ON_INDEX, pv.RULE_ROUND_HALF_AWAY, pv.RULE_ROUND_FLOOR_HALF, pv.RULE_RES_F64
RULES = [0, ON_INDEX, HALF_AWAY, FLOOR_HALF, ON_INDEX | HALF_AWAY, ON_INDEX | FLOOR_HALF, RES_F64, ON_INDEX | RES_F64]


def _unit_grid(rule, f64=True, n=5):
    """cells of size 1 starting at 0 in every dimension: the quotient IS the coordinate"""
    val = np.arange(n ** 3, dtype=np.float32).reshape(n, n, n)
    grad = np.zeros((n ** 3, 3), np.float32)
    dt = np.float64 if f64 else np.float32
    return oracle.Grid(val, grad, np.zeros(3, dt), np.full(3, n - 1, dt), np.array([[0, n - 1]] * 3, float), rule=rule)


@pytest.mark.parametrize("f64", [True, False])
def test_oracle_rules_on_hand_checked_numbers(f64):
    x = np.array([-0.75, -0.5, -0.25, 0.5, 1.5, 2.5, 3.5, 4.25, 4.5, 4.75, np.nan, np.inf], np.float32)
    pts = np.stack([x, np.full_like(x, 1.0), np.full_like(x, 2.0)], axis=1)
    expect = {  # rule -> (key of x, validity of x)
        0: ([-1, -0, -0, 0, 2, 2, 4, 4, 4, 5], [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0]),                     # half to even, value
        ON_INDEX: ([-1, -0, -0, 0, 2, 2, 4, 4, 4, 5], [0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0]),               # |q| rounds into [0, 4]
        HALF_AWAY: ([-1, -1, -0, 1, 2, 3, 4, 4, 5, 5], [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0]),
        ON_INDEX | HALF_AWAY: ([-1, -1, -0, 1, 2, 3, 4, 4, 5, 5], [0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0]),
        FLOOR_HALF: ([-1, 0, 0, 1, 2, 3, 4, 4, 5, 5], [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0]),
        ON_INDEX | FLOOR_HALF: ([-1, 0, 0, 1, 2, 3, 4, 4, 5, 5], [0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0]),
    }
    for rule, (keys, valid) in expect.items():
        key, flat, ok = oracle.voxel_index(_unit_grid(rule, f64), pts)
        assert key[:10, 0].tolist() == [int(k) for k in keys], rule
        assert ok.astype(int).tolist() == valid, rule
        assert (key[:, 1] == 1).all() and (key[:, 2] == 2).all()


@pytest.mark.parametrize("rule,f64", itertools.product(RULES, [True, False]))
def test_finalize_finds_the_float32_end_points_of_the_valid_interval(rule, f64):
    """vlo / vhi of pvamd_grid_finalize are what the query kernels' fp32 range test compares with: they must be exactly
    the first and last float32 the exact statement (here: the oracle's) calls valid, for every rule."""
    rng = [(np.float64(-0.167981), np.float64(0.202019)), (np.float64(1.0e3), np.float64(1.0e3 + 0.33)),
           (np.float64(-7.25), np.float64(-6.91))] if f64 else [(-0.167981, 0.202019), (1.0e3, 1.0e3 + 0.33), (-7.25, -6.91)]
    shape = (38, 34, 35)
    view = voxel.RangeView(rng, shape, rule=rule)
    desc = _lib.GridDesc()
    view.fill(desc)
    for d in range(3):
        desc.bb_min[d], desc.bb_max[d], desc.dbb_min[d], desc.dbb_max[d] = 0, 1, 0, 1
    desc.oob_mode = _lib.OOB_BOUNDING_BOX
    assert _lib.load().pvamd_grid_finalize(ctypes.byref(desc)) == 0 and desc.rule == rule
    rmin = (view.dmin if f64 else view.fmin).numpy()
    rmax = (view.dmax if f64 else view.fmax).numpy()
    g = oracle.Grid(np.zeros(shape, np.float32), np.zeros((np.prod(shape), 3), np.float32), rmin, rmax,
                    np.array([[0, 1]] * 3, float), index_f64=f64, rule=rule)
    if not f64:
        assert np.array_equal(np.array(g.c.fres[:]), view.fres.numpy())  # the same resolution on both sides (RES_F64 or not)
    mid = ((rmin.astype(np.float64) + rmax) / 2).astype(np.float32)
    for d in range(3):
        for end, step in ((np.float32(desc.vlo[d]), -np.inf), (np.float32(desc.vhi[d]), np.inf)):
            p_in, p_out = mid.copy(), mid.copy()
            p_in[d], p_out[d] = end, np.nextafter(end, np.float32(step))
            _, _, ok = oracle.voxel_index(g, np.stack([p_in, p_out]))
            assert ok.tolist() == [True, False], (rule, d, end)
    if rule & ON_INDEX:  # half a voxel beyond the range on each side
        res = view.dres.numpy()
        assert np.allclose(np.array(desc.vlo[:]), rmin - res / 2, atol=1e-4 * np.abs(rmin).max() + 1e-6)
        assert np.allclose(np.array(desc.vhi[:]), rmax + res / 2, atol=1e-4 * np.abs(rmax).max() + 1e-6)
    assert _lib.load().pvamd_grid_finalize(ctypes.byref(desc)) == 0
    desc.rule = HALF_AWAY | FLOOR_HALF  # two roundings at once / unknown bits are refused
    assert _lib.load().pvamd_grid_finalize(ctypes.byref(desc)) == _lib.load().pvamd_grid_finalize(ctypes.byref(desc)) < 0
    desc.rule = 1 << 9
    assert _lib.load().pvamd_grid_finalize(ctypes.byref(desc)) < 0


def _n2_float32(p, bb_min, bb_max):
    """The kernels' statements in numpy float32: t = median(p - bb_min, p - bb_max, 0) per axis, n2 = fma(tz, tz, fma(ty, ty, tx * tx))
    (the fused multiply-adds evaluated in float64 and rounded once: exact for float32 operands)."""
    p = p.astype(np.float32)
    d1, d2 = (p - bb_min.astype(np.float32)).astype(np.float32), (p - bb_max.astype(np.float32)).astype(np.float32)
    t = np.where(d1 < 0, d1, np.where(d2 > 0, d2, np.float32(0))).astype(np.float32)
    acc = (t[..., 0].astype(np.float32) * t[..., 0]).astype(np.float32)
    acc = (t[..., 1].astype(np.float64) * t[..., 1] + acc).astype(np.float32)
    return (t[..., 2].astype(np.float64) * t[..., 2] + acc).astype(np.float32)


@pytest.mark.parametrize("rule,f64", itertools.product([0, ON_INDEX], [True, False]))
def test_range_n2_bounds_every_valid_point_and_is_attained(rule, f64):
    """pvamd_grid_t.range_n2 (round 6): the composed kernels skip the range test of a leaf visit when the squared bounding-box
    distance n2 exceeds it.  That is exact only if NO valid point has a larger n2: checked here on 200,000 points of the valid
    interval [vlo, vhi] (its corners, faces and random interiors) for boxes inside, touching and sticking out of the range --
    and the bound is attained at a corner, so it is not loose either.  Host arithmetic only."""
    rng = np.random.default_rng(7 + rule + 2 * f64)
    for case in range(6):
        lo = rng.uniform(-2, 2, 3)
        ext = rng.uniform(0.2, 1.5, 3)
        ranges = [(np.float64(a), np.float64(a + e)) if f64 else (float(a), float(a + e)) for a, e in zip(lo, ext)]
        view = voxel.RangeView(ranges, (int(rng.integers(8, 40)), int(rng.integers(8, 40)), int(rng.integers(8, 40))), rule=rule)
        pad = rng.uniform(-0.05, 0.3, (3, 2)) if case else np.zeros((3, 2))  # case 0: the box IS the range; negative: it sticks out
        bb_min = (lo + pad[:, 0]).astype(np.float32)
        bb_max = np.maximum(lo + ext - pad[:, 1], bb_min + 1e-3).astype(np.float32)
        desc = _lib.GridDesc()
        view.fill(desc)
        for d in range(3):
            desc.bb_min[d], desc.bb_max[d], desc.dbb_min[d], desc.dbb_max[d] = bb_min[d], bb_max[d], float(bb_min[d]), float(bb_max[d])
        desc.oob_mode = _lib.OOB_BOUNDING_BOX
        assert _lib.load().pvamd_grid_finalize(ctypes.byref(desc)) == 0
        vlo, vhi = np.array(desc.vlo[:], np.float32), np.array(desc.vhi[:], np.float32)
        bound = np.float32(desc.range_n2)
        corners = np.array(list(itertools.product(*zip(vlo, vhi))), np.float32)
        inside = (vlo + rng.random((200_000, 3)).astype(np.float32) * (vhi - vlo)).astype(np.float32)
        inside = np.clip(inside, vlo, vhi)
        faces = inside[:20_000].copy()
        pick = rng.integers(0, 3, len(faces))
        faces[np.arange(len(faces)), pick] = np.where(rng.integers(0, 2, len(faces)) == 0, vlo[pick], vhi[pick])
        pts = np.concatenate((corners, faces, inside))
        n2 = _n2_float32(pts, bb_min, bb_max)
        assert (n2 <= bound).all(), (case, float(n2.max()), float(bound))
        assert _n2_float32(corners, bb_min, bb_max).max() == bound, (case, "the bound is a corner's n2")
    # a descriptor without a box (NaN bounds): the pre-test must never claim "out of range"
    for d in range(3):
        desc.bb_min[d] = desc.bb_max[d] = float("nan")
        desc.dbb_min[d] = desc.dbb_max[d] = float("nan")
    assert _lib.load().pvamd_grid_finalize(ctypes.byref(desc)) == 0 and desc.range_n2 == float("inf")


# ------------------------------------------------------------------------------------------------------------- GPU
def _spray(view, n, seed):
    """points at the places where the rules differ: exact half-voxel planes (and +- a few ulps), the range edges, half a voxel
    outside them, plus uniform filler"""
    g = np.random.default_rng(seed)
    mn, mx, res = view.dmin.numpy(), view.dmax.numpy(), view.dres.numpy()
    shape = np.array(view.shape)
    k = g.integers(-1, shape + 1, size=(n, 3))
    frac = g.choice([0.0, 0.5, -0.5, 0.25], size=(n, 3), p=[0.2, 0.35, 0.35, 0.1])
    pts = mn + (k + frac) * res
    pts = pts.astype(np.float32)
    nudge = g.integers(-3, 4, size=(n, 3))
    for _ in range(3):
        up = np.nextafter(pts, np.float32(np.inf))
        dn = np.nextafter(pts, np.float32(-np.inf))
        pts = np.where(nudge > 0, up, np.where(nudge < 0, dn, pts))
        nudge = nudge - np.sign(nudge)
    filler = g.uniform(mn - res, mx + res, size=(n // 4, 3)).astype(np.float32)
    return np.concatenate([pts, filler])


@pytest.fixture
def index_rule():
    prev = voxel.INDEX_RULE

    def set_rule(rule):
        voxel.INDEX_RULE = rule
    yield set_rule
    voxel.INDEX_RULE = prev


def _leaf(f64, res=0.01, padding=0.05):
    return pv.CachedSDF("rule_leaf", res, H.padded_range(H.DRILL_BB, padding, as_numpy=f64), H.drill_like_gt(), device="cuda",
                        cache_path=None)


@pytest.mark.gpu
@pytest.mark.parametrize("rule,f64", itertools.product(RULES, [True, False]))
def test_cached_kernels_follow_the_rule_bitwise(rule, f64, index_rule):
    index_rule(rule)
    cached = _leaf(f64)
    assert cached._view.rule == rule and cached._grid_desc().rule == rule
    og = H.oracle_grid_from_cached(cached)
    pts = _spray(cached._view, 60_000, seed=rule * 2 + f64)  # > 16,384: the wave-tile kernel; also a slice for the per-lane one
    for q in (pts, pts[:5001]):
        val, grad = cached(torch.from_numpy(q).cuda())
        oval, ograd, _ = oracle.cached_query(og, q)
        assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
        assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    key, flat, ok = oracle.voxel_index(og, pts)
    tp = torch.from_numpy(pts).cuda()
    assert np.array_equal(cached.voxels.get_valid_values(tp).cpu().numpy(), ok)
    assert np.array_equal(cached.voxels.ensure_index_key(tp).cpu().numpy(), key)
    out = cached.outside_surface(tp, surface_level=0.01)
    assert np.array_equal(out.cpu().numpy(), oracle.cached_outside(og, pts, 0.01))
    # float64 query points: the same rule in float64
    p64 = pts[:20_000].astype(np.float64) + np.random.default_rng(1).normal(scale=1e-12, size=(20_000, 3))
    v64, g64 = cached(torch.from_numpy(p64).cuda())
    ov, ogr, _ = oracle.cached_query_f64(og, p64)
    assert v64.dtype == torch.float64 and np.array_equal(v64.cpu().numpy(), ov, equal_nan=True)
    assert np.array_equal(g64.cpu().numpy(), ogr, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("rule", RULES[1:])
def test_composed_kernels_follow_the_rule_bitwise(rule, index_rule):
    index_rule(rule)
    S, A = 4, 24
    leaves = [_leaf(f64=(s % 2 == 0), res=0.02, padding=0.03) for s in range(S)]
    tfm = H.random_rigid(S * A, seed=rule, trans=0.1)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    # bring sprayed leaf-frame points of leaf (i mod S) under configuration 0 back to the object frame
    spray = _spray(leaves[0]._view, 24_000, seed=rule)
    inv = torch.linalg.inv(tfm.reshape(S, A, 4, 4)[:, 0].double()).numpy()
    idx = np.arange(len(spray)) % S
    pts = (np.einsum("nij,nj->ni", inv[idx, :3, :3], spray.astype(np.float64)) + inv[idx, :3, 3]).astype(np.float32)
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    oval, ograd, _ = oracle.composed_query(ogrids, tfm.numpy(), A, pts)
    dev = torch.device("cuda", torch.cuda.current_device())
    for flags in (4, 4 | 1, 2):  # wave-tile kernel with deferred / inline exact fallback, per-lane kernel
        comp._leaf_grids(dev)
        comp._query_flags = flags
        val, grad = comp(torch.from_numpy(pts).cuda())
        assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True), flags
        assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True), flags


@pytest.mark.gpu
def test_the_rules_do_differ_where_they_should(index_rule):
    """sanity of the experiment itself.  Validity on the index: other answers for points up to half a voxel outside the
    range.  The rounding rules: other answers only for quotients that are EXACTLY half-integers, which needs a grid whose
    planes float32 can hit (origin 0, resolution 2^-5) -- on the drill's ranges no float32 point ever ties, so there the
    three roundings are indistinguishable (tools/rule_exposure.py counts it).  Everywhere else: the same answers."""
    res = {}
    for rule in (0, ON_INDEX):
        index_rule(rule)
        cached = _leaf(True)
        pts = _spray(cached._view, 40_000, seed=5)
        res[rule] = cached(torch.from_numpy(pts).cuda())[0].cpu().numpy()
    assert not np.array_equal(res[0], res[ON_INDEX], equal_nan=True)
    gt = H.AnalyticEllipsoidSDF([0.5, 0.5, 0.5], [0.3, 0.2, 0.25], [[0.2, 0.8], [0.3, 0.7], [0.25, 0.75]])
    g = np.random.default_rng(2)
    ties = ((g.integers(0, 32, size=(20_000, 3)) + 0.5) / 32.0).astype(np.float32)  # exactly on the half-voxel planes
    for rule in (0, HALF_AWAY, FLOOR_HALF):
        index_rule(rule)
        c = pv.CachedSDF("pow2", 2.0 ** -5, np.array([[0.0, 1.0]] * 3), gt, device="cuda", cache_path=None)
        assert c.voxels.shape == (33, 33, 33)
        res[rule] = c.voxels.ensure_index_key(torch.from_numpy(ties).cuda()).cpu().numpy()
    q = ties.astype(np.float64) * 32
    assert np.array_equal(res[0], np.rint(q).astype(np.int64))               # half to even
    assert np.array_equal(res[HALF_AWAY], np.floor(q + 0.5).astype(np.int64))  # q > 0: away from zero = up
    assert np.array_equal(res[FLOOR_HALF], np.floor(q + 0.5).astype(np.int64))
    assert (res[0] != res[HALF_AWAY]).any()
    index_rule(0)
    cached = _leaf(True)
    lo = np.array([r[0] for r in cached.ranges]) + 0.011
    hi = np.array([r[1] for r in cached.ranges]) - 0.011
    inner = H.uniform_points(100_000, lo, hi, seed=3)  # strictly inside, almost surely off the half-voxel planes
    base = cached(inner.cuda())[0]
    for rule in RULES[1:6]:
        index_rule(rule)
        assert torch.equal(_leaf(True)(inner.cuda())[0], base)


@pytest.mark.parametrize("rule", [0, ON_INDEX, HALF_AWAY, FLOOR_HALF, ON_INDEX | HALF_AWAY, ON_INDEX | FLOOR_HALF])
def test_the_rule_of_a_given_view_is_detected(rule):
    """tools/detect_index_rule.py probes a value-range view (the real multidim_indexing one, once a maintainer has it) and
    names the rule to set; here against stand-in views that implement each rule with torch ops."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("detect_index_rule", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "detect_index_rule.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class View:
        def __init__(self, source, value_ranges=None, invalid_value=-1):
            self.shape = source.shape
            self._min = torch.tensor([min(r) for r in value_ranges])
            self._max = torch.tensor([max(r) for r in value_ranges])
            self._resolution = (self._max - self._min) / (torch.tensor(self.shape) - 1)

        def _rounded(self, key):
            q = (key - self._min) / self._resolution
            if rule & HALF_AWAY:
                return torch.sign(q) * torch.floor(torch.abs(q) + 0.5)
            if rule & FLOOR_HALF:
                return torch.floor(q + 0.5)
            return torch.round(q)

        def ensure_index_key(self, key):
            return self._rounded(key).long()

        def get_valid_values(self, key):
            if rule & ON_INDEX:
                k = self._rounded(key)
                return ((k >= 0) & (k <= torch.tensor(self.shape) - 1)).all(dim=-1)
            return ((self._min <= key) & (key <= self._max)).all(dim=-1)

    found, seen = mod.detect(View)
    assert found == rule, seen
