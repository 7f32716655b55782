"""-m gpu: the round-4 leaf loop of the wave-tile composed kernel (csrc/composed.hip: out-of-range candidates ordered by
their SQUARED norms in registers, in-range (point, leaf) pairs through a per-point LDS slot, sparse ones compacted across
visits in a wave-private queue) against the CPU oracle, bit for bit, INCLUDING the winning leaf (sdf.py:421 torch.argmin:
first minimum wins) -- and against the round-3 loop (flag 16) and the per-lane kernel (flag 2) on the same inputs."""
import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from pytorch_volumetric_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu

WAVE, LEGACY, PER_LANE, PER_LANE_LEGACY = 4, 4 | 16, 2, 2 | 16


def make_leaf(f64=True, res=0.01, padding=0.1, flip=False):
    gt = H.drill_like_gt()
    leaf = pv.CachedSDF("drill_like", res, H.padded_range(H.DRILL_BB, padding, as_numpy=f64), gt, device="cuda", cache_path=None)
    if flip:  # same values, other gradients: a wrongly picked twin leaf shows up in the gradient as well as in the leaf id
        leaf._packed[:, 1:4] = -leaf._packed[:, 1:4]
    return leaf


def query_with_leaf_ids(comp, pts, flags):
    """pvamd_composed_query through the C-ABI with out_leaf (the Python classes never ask for it)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    S = len(comp.sdfs)
    A = int(np.prod(comp.tsf_batch)) if comp.tsf_batch is not None else 1
    grids = comp._leaf_grids(dev)
    tfd = comp._tf_device(dev)
    p = pts.to(dev).contiguous()
    P = p.shape[0]
    val = torch.empty((A, P), dtype=torch.float32, device=dev)
    grad = torch.empty((A, P, 3), dtype=torch.float32, device=dev)
    leaf = torch.full((A, P), -5, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().pvamd_composed_query(_lib.ptr(grids), S, _lib.ptr(tfd), A, _lib.ptr(p), P, _lib.ptr(val),
                                                _lib.ptr(grad), _lib.ptr(leaf), flags, _lib.stream_ptr()), "pvamd_composed_query")
    torch.cuda.synchronize()
    return val.cpu().numpy(), grad.cpu().numpy(), leaf.cpu().numpy()


def check_all_kernels(leaves, tfm, A, pts):
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,) if A > 1 else None)
    oval, ograd, oleaf = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.numpy())
    for flags in (WAVE, LEGACY, PER_LANE, PER_LANE_LEGACY):
        if not (flags & 2) and pts.shape[0] < 256:
            continue
        val, grad, leaf = query_with_leaf_ids(comp, pts, flags)
        assert np.array_equal(val, oval, equal_nan=True), flags
        assert np.array_equal(grad, ograd, equal_nan=True), flags
        # a NaN value wins at its first leaf in the oracle as well; a +inf everywhere point keeps the first leaf
        assert np.array_equal(leaf, oleaf), (flags, np.argwhere(leaf != oleaf)[:5])
    return oleaf


def test_twin_leaves_tie_exactly_and_the_first_one_wins():
    """Leaves 0/1 and 2/3 are the same grid under the same transform (leaf 1 and 3 with flipped gradients): every value ties
    exactly, in range (per-point LDS slot: (value, leaf) key) and out of range (squared norms equal: the incumbent stays);
    leaf 4 alone.  torch.argmin takes the first."""
    a, b = make_leaf(), make_leaf(flip=True)
    c, d = make_leaf(f64=False, res=0.02), make_leaf(f64=False, res=0.02, flip=True)
    e = make_leaf(padding=0.05)
    A = 3
    base = H.random_rigid(3 * A, seed=21, trans=0.25).reshape(3, A, 4, 4)
    tfm = torch.stack((base[0], base[0], base[1], base[1], base[2])).reshape(5 * A, 4, 4)
    pts = H.uniform_points(40_000, [-0.6] * 3, [0.6] * 3, seed=4)
    oleaf = check_all_kernels([a, b, c, d, e], tfm, A, pts)
    assert set(np.unique(oleaf)) <= {0, 2, 4} and len(np.unique(oleaf)) == 3


def test_near_ties_between_bounding_box_distances_are_decided_by_the_rounded_roots():
    """Six copies of one leaf whose transforms differ by translations of a few 1e-8: their bounding-box distances differ by
    0..3 ulps, so squared norms that differ may still have the SAME correctly rounded root (then the earlier leaf wins) --
    the band the kernel re-decides with both exact roots.  Also copies in DEcreasing order of distance, where every later
    leaf is a little closer."""
    leaf = make_leaf()
    A = 2
    base = H.random_rigid(A, seed=5, trans=0.2)
    for sign in (1.0, -1.0):
        stack = []
        for s in range(6):
            m = base.clone()
            m[:, :3, 3] += sign * s * torch.tensor([3e-8, -2e-8, 1e-8])
            stack.append(m)
        tfm = torch.stack(stack).reshape(6 * A, 4, 4)
        pts = H.uniform_points(30_000, [-1.5] * 3, [1.5] * 3, seed=9)  # mostly out of every range
        oleaf = check_all_kernels([leaf] * 6, tfm, A, pts)
        assert len(np.unique(oleaf)) >= 3  # the order does get decided differently from point to point


def test_dense_and_sparse_in_range_visits_and_a_queue_that_fills():
    """Points drawn INSIDE the leaf ranges in clusters (whole waves in range: the owning lanes look their records up and go
    through the LDS slot directly), the same points shuffled (a few lanes per visit: the queue, which fills and drains
    inside a pass when 12 leaves overlap), and a scattered cloud -- one call."""
    S, A = 12, 4
    leaves = [make_leaf(f64=(s % 2 == 0), res=0.02 if s % 3 else 0.01, padding=0.15) for s in range(S)]
    tfm = H.random_rigid(S * A, seed=31, trans=0.12)  # heavily overlapping ranges
    g = np.random.default_rng(2)
    centres = g.uniform(-0.25, 0.25, size=(64, 3)).repeat(256, axis=0)
    clustered = (centres + g.normal(scale=0.01, size=centres.shape)).astype(np.float32)
    shuffled = clustered[g.permutation(len(clustered))]
    cloud = g.uniform(-0.7, 0.7, size=(16_384 + 77, 3)).astype(np.float32)
    pts = torch.from_numpy(np.concatenate((clustered, shuffled, cloud)))
    oleaf = check_all_kernels(leaves, tfm, A, pts)
    assert len(np.unique(oleaf)) == S


def test_shaky_index_estimates_inside_dense_and_queued_visits():
    """Query points on the half-voxel planes of a leaf far from its origin (the fp32 index estimate cannot be trusted there):
    in dense visits the owning lane flags its point and redoes it exactly after the loop, in the drained queue the exact
    statements run inline -- both must give the reference index."""
    gt = H.AnalyticEllipsoidSDF([7.0, -5.0, 3.0], [0.3, 0.2, 0.25], [[6.7, 7.3], [-5.2, -4.8], [2.75, 3.25]])
    rng = [(6.5, 7.5), (-5.5, -4.5), (2.5, 3.5)]
    leaves = [pv.CachedSDF(f"far{i}", 0.02, np.array(rng) if i % 2 == 0 else rng, gt, device="cuda", cache_path=None)
              for i in range(4)]
    S, A = 4, 6
    tfm = H.random_rigid(S * A, seed=11, trans=0.2)
    v = leaves[0]._view
    g = np.random.default_rng(5)
    n = 1 << 14
    k = g.integers(0, np.array(v.shape) - 1, size=(n, 3))
    on_plane = v.dmin.numpy() + (k + 0.5) * v.dres.numpy() + g.normal(scale=2e-6, size=(n, 3))
    inv = torch.linalg.inv(tfm.reshape(S, A, 4, 4)[:, 0].double()).numpy()
    # sorted by target leaf: runs of 4096 points land in ONE leaf's range (dense visits); then the same points interleaved
    by_leaf = np.concatenate([np.stack([inv[s, :3, :3] @ q + inv[s, :3, 3] for q in on_plane[s::S]]) for s in range(S)])
    mixed = by_leaf[g.permutation(len(by_leaf))]
    far = g.uniform(-9, 9, size=(4096, 3))
    pts = torch.from_numpy(np.concatenate((by_leaf, mixed, far)).astype(np.float32))
    check_all_kernels(leaves, tfm, A, pts)


def test_nan_and_infinite_values_in_points_transforms_and_caches():
    """NaN counts as the minimum and the FIRST NaN wins (sdf.py:421): a NaN stored in a cache (in-range slot key 0), a NaN
    coming out of a transform (out-of-range register minimum), infinite coordinates (every leaf answers +inf: the first
    visited leaf stays), -0.0 against +0.0 cached values (equal: the first wins)."""
    a, b, c = make_leaf(), make_leaf(f64=False), make_leaf()
    b._packed[::7, 0] = float("nan")
    a._packed[::5, 0] = 0.0
    c._packed[::5, 0] = -0.0
    c._packed[:, 1:4] = -c._packed[:, 1:4]
    A = 2
    base = H.random_rigid(2 * A, seed=3, trans=0.05).reshape(2, A, 4, 4)
    tfm = torch.stack((base[0], base[1], base[0])).reshape(3 * A, 4, 4).clone()
    pts = H.uniform_points(20_000, [-0.3] * 3, [0.3] * 3, seed=8)
    pts[17] = torch.tensor([float("nan"), 0.0, 0.1])
    pts[300] = torch.tensor([float("inf"), float("inf"), float("inf")])
    pts[301] = torch.tensor([0.0, float("-inf"), 0.0])
    check_all_kernels([a, b, c], tfm, A, pts)
    tfm[1 * A + 1, 0, 3] = float("nan")  # leaf 1 of configuration 1: every coordinate x is NaN
    check_all_kernels([a, b, c], tfm, A, pts)
