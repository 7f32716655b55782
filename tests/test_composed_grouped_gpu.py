"""-m gpu: the chunk-grouped composed query (round 6: pvamd_group_points + pvamd_composed_query_grouped) -- points regrouped
spatially inside chunks of consecutive caller points, results restored to the caller's order through LDS -- against the CPU
oracle and against the ungrouped kernels, bit for bit (sdf.py:392-433 of the reference)."""
import ctypes

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from pytorch_volumetric_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def make_leaf(f64=True, res=0.01, padding=0.1):
    gt = H.drill_like_gt()
    return pv.CachedSDF("drill_like", res, H.padded_range(H.DRILL_BB, padding, as_numpy=f64), gt, device="cuda",
                        cache_path=None)


def scene_points(n, seed, extent=0.5):
    return H.uniform_points(n, [-extent] * 3, [extent] * 3, seed)


def composed(S, A, seed, trans=0.3):
    leaves = [make_leaf(f64=(s % 2 == 0)) for s in range(S)]
    tfm = H.random_rigid(S * A, seed=seed, trans=trans)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    return comp, leaves, tfm


def same_bits(a, b):
    return np.array_equal(a.view(np.int32), b.view(np.int32))


@pytest.mark.parametrize("S,A,P", [(8, 3, 4096), (8, 2, 4097), (8, 5, 12_345), (3, 7, 8191), (1, 2, 4096), (20, 2, 5000), (64, 2, 4500), (70, 1, 4096)])
def test_grouped_matches_the_oracle_and_the_ungrouped_kernels_bitwise(S, A, P):
    """Any P >= one chunk (a ragged end: the last chunk is moved back and overlaps its neighbour), mixed float64 / float32 index
    arithmetic per leaf, odd row starts of the (A, P) outputs."""
    chunk = _lib.group_chunk_points()
    assert P >= chunk
    comp, leaves, tfm = composed(S, A, seed=S * 100 + A)
    pts = scene_points(P, seed=P, extent=0.6).cuda()
    comp.group_points = True
    val, grad = comp(pts)
    comp.group_points = False
    val0, grad0 = comp(pts)
    assert val.shape == (A, P) and grad.shape == (A, P, 3)
    assert same_bits(val.cpu().numpy(), val0.cpu().numpy()) and same_bits(grad.cpu().numpy(), grad0.cpu().numpy())
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def test_grouped_leaf_ids_and_a_single_configuration_through_the_c_abi():
    """out_leaf (the arg-min leaf of every point, written to the caller's order from inside the leaf loop) and A = 1, straight
    through the two entry points."""
    S, A, P = 8, 1, 9000
    comp, leaves, tfm = composed(S, A, seed=5)
    pts = scene_points(P, seed=3, extent=0.6).cuda()
    lib = _lib.load()
    dev = pts.device
    grids = comp._leaf_grids(dev)
    scratch = _lib.group_points(pts)
    val = torch.empty((A, P), device=dev)
    grad = torch.empty((A, P, 3), device=dev)
    leaf = torch.full((A, P), -1, dtype=torch.int32, device=dev)
    _lib.check(lib.pvamd_composed_query_grouped(_lib.ptr(grids), S, _lib.ptr(comp._tf_device(dev)), A, _lib.ptr(scratch), P,
                                                _lib.ptr(val), _lib.ptr(grad), _lib.ptr(leaf), 0, _lib.stream_ptr()), "grouped")
    oval, ograd, oleaf = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    assert np.array_equal(leaf.cpu().numpy(), oleaf)


def test_grouped_refuses_what_it_does_not_cover():
    lib = _lib.load()
    chunk = _lib.group_chunk_points()
    assert lib.pvamd_group_scratch_bytes(chunk - 1) == 0
    assert lib.pvamd_group_scratch_bytes(chunk) > chunk * 14
    comp, _, _ = composed(2, 2, seed=1)
    pts = scene_points(chunk, seed=1).cuda()
    scratch = _lib.group_points(pts)
    grids = comp._leaf_grids(pts.device)
    val = torch.empty((2, chunk), device="cuda")
    grad = torch.empty((2, chunk, 3), device="cuda")
    args = (_lib.ptr(grids), 2, _lib.ptr(comp._tf_device(pts.device)), 2, _lib.ptr(scratch))
    assert lib.pvamd_composed_query_grouped(*args, chunk - 1, _lib.ptr(val), _lib.ptr(grad), None, 0, _lib.stream_ptr()) == _lib.E_SHAPE
    assert lib.pvamd_composed_query_grouped(*args, chunk, _lib.ptr(val), _lib.ptr(grad), None, _lib.COMPOSED_INLINE_EXACT,
                                            _lib.stream_ptr()) == -4
    assert lib.pvamd_group_points(_lib.ptr(pts), chunk - 1, _lib.ptr(scratch), _lib.stream_ptr()) == _lib.E_SHAPE
    assert lib.pvamd_group_points(None, chunk, _lib.ptr(scratch), _lib.stream_ptr()) == -1
    torch.cuda.synchronize()


def test_grouped_with_nan_infinite_and_coincident_points():
    """Non-finite coordinates take part in no bound (the run they are in visits every leaf) and land in some Hilbert cell; a
    chunk of identical points has a zero-size box.  Results as the oracle's, NaN for NaN."""
    S, A = 6, 3
    chunk = _lib.group_chunk_points()
    P = 3 * chunk
    comp, leaves, tfm = composed(S, A, seed=9)
    pts = scene_points(P, seed=11, extent=0.6)
    pts[5] = float("nan")
    pts[17, 1] = float("inf")
    pts[chunk + 3, 2] = -float("inf")
    pts[chunk + 100, 0] = float("nan")
    pts[2 * chunk:] = pts[2 * chunk]  # the third chunk: one point, repeated
    pts = pts.cuda()
    comp.group_points = True
    val, grad = comp(pts)
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def test_grouped_on_compact_points_uses_the_leaf_mask_and_stays_exact():
    """Points inside a small region (every run's sphere is far smaller than the scene: the masked leaf loop with its
    refinement runs) and leaves spread far apart, so that whole leaves are dropped per run."""
    S, A = 8, 4
    chunk = _lib.group_chunk_points()
    P = 2 * chunk + 77
    comp, leaves, tfm = composed(S, A, seed=21, trans=1.5)
    pts = (scene_points(P, seed=2, extent=0.08) + torch.tensor([0.3, -0.2, 0.1])).cuda()
    comp.group_points = True
    val, grad = comp(pts)
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def test_query_into_grouped_at_odd_addresses_and_replayed_from_a_graph():
    """query_into picks the grouped kernel on its own for a batch in the wave-tile regime; buffers that are only dword
    aligned; the two launches captured in a hipGraph and replayed after the transforms changed in place."""
    S, A = 8, 40
    P = 2 * _lib.group_chunk_points() + 257 * 4 + 1  # A * tiles >= 32768 needs P >= 209,716 at A = 40 ... not here: force it
    comp, leaves, tfm = composed(S, A, seed=4)
    comp.group_points = True
    pts = scene_points(P, seed=8, extent=0.6)
    pbuf = torch.zeros(3 * P + 8, device="cuda")
    pbuf[1:1 + 3 * P] = pts.reshape(-1).cuda()
    pview = pbuf[1:1 + 3 * P].view(P, 3)
    vbuf = torch.full((A * P + 8,), -7.0, device="cuda")
    gbuf = torch.full((3 * A * P + 8,), -7.0, device="cuda")
    vview, gview = vbuf[3:3 + A * P].view(A, P), gbuf[1:1 + 3 * A * P].view(A, P, 3)
    comp.query_into(pview, vview, gview)
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    oval, ograd, _ = oracle.composed_query(ogrids, tfm.numpy(), A, pts.numpy())
    assert np.array_equal(vview.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(gview.cpu().numpy(), ograd, equal_nan=True)
    assert (vbuf[:3] == -7).all() and (vbuf[3 + A * P:] == -7).all() and (gbuf[:1] == -7).all() and (gbuf[1 + 3 * A * P:] == -7).all()
    # graph replay with new transforms written in place
    tf_dev = comp._tf_device(pview.device)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        comp.query_into(pview, vview, gview)
        with torch.cuda.graph(g, stream=s):
            comp.query_into(pview, vview, gview)
    tfm2 = H.random_rigid(S * A, seed=44, trans=0.3)
    tf_dev.copy_(tfm2.cuda())
    vbuf.fill_(-7.0)
    g.replay()
    torch.cuda.synchronize()
    oval, ograd, _ = oracle.composed_query(ogrids, tfm2.numpy(), A, pts.numpy())
    assert np.array_equal(vview.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(gview.cpu().numpy(), ograd, equal_nan=True)


def test_a_captured_graph_keeps_its_scratch_while_other_sizes_come_and_go():
    """query_into remembers one scratch buffer per point count (8 at most) -- but a buffer a stream capture has seen belongs to
    a graph that replays with its address: ten other sizes later the first graph still answers correctly."""
    S, A = 4, 3
    chunk = _lib.group_chunk_points()
    comp, leaves, tfm = composed(S, A, seed=9)
    comp.group_points = True
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    P0 = chunk + 33
    pts0 = scene_points(P0, seed=3, extent=0.6).cuda()
    val0, grad0 = torch.empty((A, P0), device="cuda"), torch.empty((A, P0, 3), device="cuda")
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        comp.query_into(pts0, val0, grad0)
        with torch.cuda.graph(g, stream=side):
            comp.query_into(pts0, val0, grad0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):  # the other sizes on the SAME stream: same table keys apart from the size
        for i in range(10):
            P = chunk + 100 * (i + 1)
            p = scene_points(P, seed=20 + i, extent=0.6).cuda()
            v, gr = torch.empty((A, P), device="cuda"), torch.empty((A, P, 3), device="cuda")
            comp.query_into(p, v, gr)
            v.fill_(float("nan"))  # and the freed buffers get reused by these allocations
    torch.cuda.synchronize()
    table = comp.__dict__["_group_scratch"]
    assert sum(1 for e in table.values() if e[1]) == 1 and sum(1 for e in table.values() if not e[1]) <= 8
    val0.fill_(-1.0)
    g.replay()
    torch.cuda.synchronize()
    oval, ograd, _ = oracle.composed_query(ogrids, tfm.numpy(), A, pts0.cpu().numpy())
    assert np.array_equal(val0.cpu().numpy(), oval, equal_nan=True) and np.array_equal(grad0.cpu().numpy(), ograd, equal_nan=True)


def test_auto_takes_the_grouped_kernel_only_in_its_regime():
    comp, _, _ = composed(2, 2, seed=1)
    chunk = _lib.group_chunk_points()
    assert comp.group_points == "auto"
    assert not comp._grouping_pays(1, 1 << 22, 0)            # a single configuration cannot amortise the sort pass
    assert not comp._grouping_pays(200, chunk - 1, 0)        # less than a chunk
    assert not comp._grouping_pays(200, 15_251, 0)           # too few tiles to fill the chip: the per-lane kernel's regime
    assert comp._grouping_pays(200, 1 << 18, 0)              # C4
    assert not comp._grouping_pays(200, 1 << 18, _lib.COMPOSED_INLINE_EXACT)  # gather-bound grids: the bucketed path's regime


@pytest.mark.parametrize("P,A,world", [(262_144 // 4, 24, 8), (40_001, 10, 8), (5000, 3, 3)])
def test_eight_shards_computed_one_by_one_unpack_to_the_unsharded_call(P, A, world):
    """the packed one-collective path of dist.ShardedSDF, with the all-gather's result assembled
    by hand -- every rank's slab of packed records computed separately on this one GPU (rank r's slice of the points, padded
    to whole tiles as ShardedSDF._packed_call pads it; the chunk-grouped packed kernel where the slice is large enough, the
    wave-tile packed kernel otherwise), laid out (world, A, Pp, 4) as all_gather_into_tensor leaves it -- through
    dist.packed_index + pvamd_unpack_records: equal, bit for bit, to the unsharded call.  No process group."""
    from pytorch_volumetric_amd import dist as pvdist
    S = 8
    comp, leaves, tfm = composed(S, A, seed=31)
    pts = scene_points(P, seed=12, extent=0.6).cuda()
    comp.group_points = True  # (refused by itself below one chunk: _grouping_pays)
    dv, dg = comp(pts)
    _, _, chunk = pvdist.shard_range(P, world, 0)
    Pp = -(-chunk // 256) * 256
    gathered = torch.empty((world, A, Pp, 4), device="cuda")
    used_grouped = False
    for r in range(world):
        start, stop, _ = pvdist.shard_range(P, world, r)
        mine = pts[start:stop]
        if mine.shape[0] < Pp:
            mine = torch.cat((mine, pts[:1].expand(Pp - mine.shape[0], 3)), dim=0)
        used_grouped |= comp._grouping_pays(A, Pp, 0)
        gathered[r] = comp.query_packed(mine.contiguous())
    assert used_grouped == (Pp >= _lib.group_chunk_points())
    index = pvdist.packed_index(P, chunk, A, Pp, pts.device)
    sh = pvdist.ShardedSDF(comp)
    val, grad = sh._unpack_records(gathered, index, P, Pp, A, pts.device)
    assert same_bits(val.cpu().numpy(), dv.cpu().numpy()) and same_bits(grad.cpu().numpy(), dg.cpu().numpy())


@pytest.mark.parametrize("S,A,P", [(8, 1, 4096), (8, 1, 50_001), (5, 3, 9000), (64, 1, 4200), (70, 2, 4100)])
def test_fused_in_workgroup_sort_matches_the_oracle_bitwise(S, A, P):
    """composed_query_fused (round 6): pvamd_composed_query itself regroups every chunk inside its workgroup -- no scratch, no
    pre-pass -- for a single configuration too (C3).  Forced here at sizes the entry point would leave to the older kernels; more
    leaves than the culling table holds; out_leaf; against the oracle and against the ungrouped kernels."""
    comp, leaves, tfm = composed(S, A, seed=7 * S + A)
    pts = scene_points(P, seed=P + 1, extent=0.6)
    pts[11] = float("nan")
    pts[P - 1, 2] = float("inf")
    pts = pts.cuda()
    lib = _lib.load()
    dev = pts.device
    grids = comp._leaf_grids(dev)
    out = {}
    for flags in (_lib.COMPOSED_FORCE_FUSED, _lib.COMPOSED_NO_GROUPING):
        val = torch.empty((A, P), device=dev)
        grad = torch.empty((A, P, 3), device=dev)
        leaf = torch.full((A, P), -1, dtype=torch.int32, device=dev)
        _lib.check(lib.pvamd_composed_query(_lib.ptr(grids), S, _lib.ptr(comp._tf_device(dev)), A, _lib.ptr(pts), P, _lib.ptr(val),
                                            _lib.ptr(grad), _lib.ptr(leaf), flags, _lib.stream_ptr()), "pvamd_composed_query")
        out[flags] = (val.cpu().numpy(), grad.cpu().numpy(), leaf.cpu().numpy())
    f, n = out[_lib.COMPOSED_FORCE_FUSED], out[_lib.COMPOSED_NO_GROUPING]
    assert same_bits(f[0], n[0]) and same_bits(f[1], n[1]) and np.array_equal(f[2], n[2])
    oval, ograd, oleaf = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.cpu().numpy())
    assert np.array_equal(f[0], oval, equal_nan=True) and np.array_equal(f[1], ograd, equal_nan=True)
    assert np.array_equal(f[2], oleaf)


def test_the_entry_point_picks_the_fused_kernel_for_a_large_single_configuration():
    """A = 1, 4,194,304 + 77 points (C3's size, ragged): what comp(points) runs is the fused kernel; same bits as with
    group_points = False (the per-lane kernel) and as the oracle on three slices."""
    comp, leaves, tfm = composed(8, 1, seed=3)
    comp.set_transforms(pv.Transform3d(matrix=tfm))  # no batch: flat outputs
    P = (1 << 22) + 77
    pts = H.uniform_points(P, [-0.5] * 3, [0.5] * 3, seed=5).cuda()
    v1, g1 = comp(pts)
    comp.group_points = False
    v0, g0 = comp(pts)
    assert v1.shape == (P,) and torch.equal(v1.view(torch.int32), v0.view(torch.int32)) and torch.equal(g1.view(torch.int32), g0.view(torch.int32))
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    for a, b in ((0, 3000), (P // 2, P // 2 + 3000), (P - 3000, P)):
        oval, ograd, _ = oracle.composed_query(ogrids, tfm.numpy(), 1, pts[a:b].cpu().numpy())
        assert np.array_equal(v1[a:b].cpu().numpy(), oval[0], equal_nan=True)
        assert np.array_equal(g1[a:b].cpu().numpy(), ograd[0], equal_nan=True)


def test_fused_kernel_on_points_that_are_already_ordered():
    """A chunk whose 256-point tiles each span less than half of it (an ordered slice, a pre-sorted set) is not sorted inside the
    workgroup (composed_query_fused: identity positions through the same leaf loop); a set that is ordered in its first chunks
    and scattered in its last ones takes both paths in one launch.  The oracle's bits either way."""
    comp, leaves, tfm = composed(8, 1, seed=13)
    ax = torch.linspace(-0.6, 0.6, 128)
    slab = torch.cartesian_prod(ax, ax, torch.tensor([0.05]))              # 16,384 points of a planar slice, row by row
    pts = torch.cat((slab, scene_points(3 * 4096 + 5, seed=1, extent=0.6))).cuda().contiguous()
    P = pts.shape[0]
    lib = _lib.load()
    dev = pts.device
    grids = comp._leaf_grids(dev)
    out = {}
    for flags in (_lib.COMPOSED_FORCE_FUSED, _lib.COMPOSED_NO_GROUPING):
        val = torch.empty((1, P), device=dev)
        grad = torch.empty((1, P, 3), device=dev)
        _lib.check(lib.pvamd_composed_query(_lib.ptr(grids), 8, _lib.ptr(comp._tf_device(dev)), 1, _lib.ptr(pts), P, _lib.ptr(val),
                                            _lib.ptr(grad), None, flags, _lib.stream_ptr()), "pvamd_composed_query")
        out[flags] = (val.cpu().numpy(), grad.cpu().numpy())
    f, n = out[_lib.COMPOSED_FORCE_FUSED], out[_lib.COMPOSED_NO_GROUPING]
    assert same_bits(f[0], n[0]) and same_bits(f[1], n[1])
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), 1, pts.cpu().numpy())
    assert np.array_equal(f[0], oval, equal_nan=True) and np.array_equal(f[1], ograd, equal_nan=True)
