"""-m gpu: CachedSDF HIP path (through the C ABI) vs the CPU oracle, bit-exact (BASELINE config C2)."""
import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def make_cached(resolution=0.01, padding=0.1, f64=True, oob=pv.OutOfBoundsStrategy.BOUNDING_BOX, device="cuda"):
    gt = H.drill_like_gt()
    rng = H.padded_range(H.DRILL_BB, padding, as_numpy=f64)
    return pv.CachedSDF("drill_like", resolution, rng, gt, out_of_bounds_strategy=oob, device=device, cache_path=None)


def query_points(cached, n, seed, margin=0.05):
    lo = np.array([r[0] for r in cached.ranges]) - margin
    hi = np.array([r[1] for r in cached.ranges]) + margin
    return H.uniform_points(n, lo, hi, seed)


@pytest.mark.parametrize("f64", [True, False])
def test_grid_shape_is_the_c2_grid(f64):
    c = make_cached(f64=f64)
    assert c._view.shape == (37, 33, 40)  # drill, res 0.01, pad 0.1
    assert c._view.index_f64 == f64


@pytest.mark.parametrize("f64", [True, False])
def test_voxel_index_bit_exact(f64):
    c = make_cached(f64=f64)
    og = H.oracle_grid_from_cached(c)
    pts = query_points(c, 200_003, seed=1)
    key = c.voxels.ensure_index_key(pts.cuda())
    valid = c.voxels.get_valid_values(pts.cuda())
    okey, oflat, ovalid = oracle.voxel_index(og, pts.numpy())
    assert np.array_equal(key.cpu().numpy(), okey)
    assert np.array_equal(valid.cpu().numpy(), ovalid)
    flat = c.voxels.ravel_multi_index(key, c.voxels.shape)
    assert np.array_equal(flat.cpu().numpy(), oflat)
    assert 0.05 < (~ovalid).mean() < 0.95  # the sample exercises both branches


@pytest.mark.parametrize("f64", [True, False])
def test_half_voxel_boundaries_round_half_even(f64):
    """Points placed on and next to the half-voxel planes and the range edges: the rounding mode and the inclusive
    range test are where index arithmetic can go wrong."""
    c = make_cached(f64=f64)
    og = H.oracle_grid_from_cached(c)
    v = c._view
    mn = (v.dmin if f64 else v.fmin).double().numpy()
    res = (v.dres if f64 else v.fres).double().numpy()
    mx = (v.dmax if f64 else v.fmax).double().numpy()
    ks = np.arange(0, 33)
    base = []
    for off in (0.5, 0.5 - 1e-7, 0.5 + 1e-7, 0.0, 1.0):
        base.append(mn[None, :] + (ks[:, None] + off) * res[None, :])
    pts = np.concatenate(base + [mn[None], mx[None], np.nextafter(mn.astype(np.float32), -1)[None].astype(np.float64),
                                 np.nextafter(mx.astype(np.float32), 10)[None].astype(np.float64)]).astype(np.float32)
    key = c.voxels.ensure_index_key(torch.from_numpy(pts).cuda()).cpu().numpy()
    valid = c.voxels.get_valid_values(torch.from_numpy(pts).cuda()).cpu().numpy()
    okey, _, ovalid = oracle.voxel_index(og, pts)
    assert np.array_equal(key, okey)
    assert np.array_equal(valid, ovalid)


@pytest.mark.parametrize("f64", [True, False])
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 100_001])
def test_cached_query_matches_oracle_bitwise(f64, n):
    c = make_cached(f64=f64)
    og = H.oracle_grid_from_cached(c)
    pts = query_points(c, n, seed=n + 7)
    val, grad = c(pts.cuda())
    assert val.shape == (n,) and grad.shape == (n, 3) and val.device.type == "cuda"
    oval, ograd, _ = oracle.cached_query(og, pts.numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def test_misaligned_and_strided_inputs_take_the_scalar_path():
    c = make_cached()
    og = H.oracle_grid_from_cached(c)
    big = query_points(c, 4099, seed=3).cuda()
    sl = big[1:]  # data_ptr offset by 12 B: not 16-byte aligned
    val, grad = c(sl)
    oval, ograd, _ = oracle.cached_query(og, sl.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    strided = big[::2]
    val, grad = c(strided)
    oval, ograd, _ = oracle.cached_query(og, strided.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)


def test_batch_dims_dtype_and_device_follow_the_reference():
    c = make_cached(device="cuda")
    pts = query_points(c, 6 * 50, seed=5).reshape(2, 3, 50, 3)
    val, grad = c(pts.cuda())
    assert val.shape == (2, 3, 50) and grad.shape == (2, 3, 50, 3)
    v64, g64 = c(pts.double().cuda())
    assert v64.dtype == torch.float64 and g64.dtype == torch.float64  # output dtype = query dtype (sdf.py:545-547)
    c_cpu = make_cached(device="cpu")
    v, g = c_cpu(pts)  # cpu in, self.device out
    assert v.device.type == "cpu" and torch.equal(v, val.cpu())


def test_nan_and_inf_points_are_out_of_bounds():
    c = make_cached()
    og = H.oracle_grid_from_cached(c)
    pts = query_points(c, 64, seed=9)
    pts[3, 1] = float("nan")
    pts[10, 0] = float("inf")
    pts[11, 2] = -float("inf")
    val, grad = c(pts.cuda())
    oval, ograd, ooob = oracle.cached_query(og, pts.numpy())
    assert ooob[3] and ooob[10] and ooob[11]
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def test_voxel_centres_map_to_themselves():
    """The invariant the reference asserts under debug_check_sdf (sdf.py:508-512)."""
    for f64 in (False, True):
        c = make_cached(f64=f64)
        coords, centres = pv.get_coordinates_and_points_in_grid(c.resolution, c.ranges)
        val, grad = c(centres.cuda())
        packed = c._packed
        valid = c.voxels.get_valid_values(centres.cuda())
        # fp32 centre coordinates on the boundary planes can round to just outside a float64 range (and then take the
        # out-of-range branch, as they would in the reference); every interior centre must be in range
        idx = torch.cartesian_prod(*[torch.arange(len(x)) for x in coords])
        interior = ((idx > 0) & (idx < torch.tensor(c._view.shape) - 1)).all(dim=-1).cuda()
        assert valid[interior].all()
        assert torch.equal(val[valid], packed[valid, 0])
        assert torch.equal(grad[valid], packed[valid, 1:4])
        key = c.voxels.ensure_index_key(centres.cuda())
        assert torch.equal(key.cpu(), idx)  # centre i -> index i, in range or not


def test_bounding_box_fallback_properties():
    """sdf.py:574-582: the fallback under-approximates the true distance and points away from the box."""
    c = make_cached()
    lo = np.array([r[0] for r in c.ranges]) - 0.3
    hi = np.array([r[1] for r in c.ranges]) + 0.3
    pts = H.uniform_points(50_000, lo, hi, seed=11).cuda()
    val, grad = c(pts)
    oob = ~c.voxels.get_valid_values(pts)
    bb = c.bb.float()
    q = torch.maximum(bb[:, 0] - pts[oob], pts[oob] - bb[:, 1]).clamp(min=0)
    assert torch.allclose(val[oob], q.norm(dim=-1), atol=1e-6)
    assert torch.allclose(grad[oob].norm(dim=-1), torch.ones_like(val[oob]), atol=1e-5)
    assert (val[oob] > 0).all()


def test_lookup_gt_sdf_strategy_queries_ground_truth_out_of_range():
    gt = H.drill_like_gt()
    c = make_cached(oob=pv.OutOfBoundsStrategy.LOOKUP_GT_SDF)
    pts = query_points(c, 10_000, seed=13).cuda()
    val, grad = c(pts)
    oob = ~c.voxels.get_valid_values(pts)
    gv, gg = gt(pts[oob])
    assert torch.equal(val[oob], gv) and torch.equal(grad[oob], gg)
    og = H.oracle_grid_from_cached(c, oob_mode=0)
    oval, ograd, ooob = oracle.cached_query(og, pts.cpu().numpy())
    inb = ~ooob
    assert np.array_equal(val.cpu().numpy()[inb], oval[inb])
    assert np.array_equal(oob.cpu().numpy(), ooob)


def test_outside_surface_matches_oracle():
    c = make_cached()
    og = H.oracle_grid_from_cached(c)
    pts = query_points(c, 30_001, seed=17)
    for level in (0.0, 0.01, -0.005):
        out = c.outside_surface(pts.cuda(), surface_level=level)
        assert np.array_equal(out.cpu().numpy(), oracle.cached_outside(og, pts.numpy(), level))


def test_full_size_c2_properties():
    """BASELINE C2 at full size (1M points): parity through size-independent properties -- determinism, agreement of
    the vector and scalar code paths, and permutation equivariance."""
    c = make_cached()
    pts = query_points(c, 1_000_000, seed=21).cuda()
    v1, g1 = c(pts)
    v2, g2 = c(pts)
    assert torch.equal(v1, v2) and torch.equal(g1.nan_to_num(7.0), g2.nan_to_num(7.0))
    shifted = torch.cat((pts[:1], pts))[1:]  # same values at a 12-byte-shifted address: scalar kernel
    v3, g3 = c(shifted)
    assert torch.equal(v1, v3) and torch.equal(g1.nan_to_num(7.0), g3.nan_to_num(7.0))
    perm = torch.randperm(pts.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    v4, _ = c(pts[perm])
    assert torch.equal(v1[perm], v4)
    # and a 100k-point slice against the oracle
    og = H.oracle_grid_from_cached(c)
    oval, ograd, _ = oracle.cached_query(og, pts[:100_000].cpu().numpy())
    assert np.array_equal(v1[:100_000].cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(g1[:100_000].cpu().numpy(), ograd, equal_nan=True)


@pytest.mark.parametrize("f64", [True, False])
def test_index_fast_path_agrees_with_exact_division_at_rounding_boundaries(f64):
    """The query kernels estimate the index with a multiply and fall back to the reference's exact division only near
    half-voxel planes.  Hammer those planes: values must still match the oracle (which always divides) bit for bit."""
    c = make_cached(f64=f64)
    og = H.oracle_grid_from_cached(c)
    v = c._view
    mn = (v.dmin if f64 else v.fmin).double().numpy()
    res = (v.dres if f64 else v.fres).double().numpy()
    rng = np.random.default_rng(0)
    n = 400_000
    k = rng.integers(0, np.array(v.shape) - 1, size=(n, 3))
    # exactly on, and within a few float32 ulps of, the half-voxel planes (in all three coordinates at once)
    ulps = rng.integers(-6, 7, size=(n, 3))
    base = (mn[None, :] + (k + 0.5) * res[None, :]).astype(np.float32)
    pts = base.copy()
    for _ in range(6):
        up = np.nextafter(pts, np.float32(np.inf))
        dn = np.nextafter(pts, np.float32(-np.inf))
        pts = np.where(ulps > 0, up, np.where(ulps < 0, dn, pts))
        ulps = ulps - np.sign(ulps)
    val, grad = c(torch.from_numpy(pts).cuda())
    oval, ograd, ooob = oracle.cached_query(og, pts)
    assert not ooob.any()
    assert np.array_equal(val.cpu().numpy(), oval) and np.array_equal(grad.cpu().numpy(), ograd)
    key = c.voxels.ensure_index_key(torch.from_numpy(pts).cuda()).cpu().numpy()
    okey, _, _ = oracle.voxel_index(og, pts)
    assert np.array_equal(key, okey)
    # the sample really straddles the rounding boundary: both neighbours occur
    assert ((okey == k).any() and (okey == k + 1).any())


class CircleSDF(pv.ObjectFrameSDF):
    """Planar ground truth: a circle of radius r at the origin (what SphereSDF is in two dimensions)."""

    def __init__(self, r):
        self.r = r

    def __call__(self, p):
        n = torch.linalg.norm(p, dim=-1)
        return n - self.r, p / (n.unsqueeze(-1) + 1e-12)

    def surface_bounding_box(self, padding=0., padding_ratio=0.):
        e = self.r + padding + padding_ratio * self.r
        return torch.tensor([[-e, e], [-e, e]])


@pytest.mark.parametrize("numpy_range", [True, False])
def test_planar_cache_two_dimensional_points(numpy_range):
    """ObjectFrameSDF is d-dimensional, d = 2 or 3 (sdf.py:222).  A planar CachedSDF answers (..., 2) points with (...,)
    values and (..., 2) gradients: in range the nearest cell of the planar grid (x slowest, y fastest), out of range the
    planar bounding-box statements of sdf.py:559-571."""
    gt = CircleSDF(0.3)
    rng = np.array([[-0.5, 0.5], [-0.4, 0.6]]) if numpy_range else [(-0.5, 0.5), (-0.4, 0.6)]
    c = pv.CachedSDF("circle", 0.05, rng, gt, device="cuda", cache_path=None)
    assert c.voxels.shape == (21, 21) and c._view.index_f64 == numpy_range
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(3, 5000, 2, generator=g) * 1.6 - 0.8
    val, grad = c(pts.cuda())
    assert val.shape == (3, 5000) and grad.shape == (3, 5000, 2)
    p = pts.reshape(-1, 2).double().numpy()
    lo = np.array([r[0] for r in c.ranges], dtype=np.float64)
    hi = np.array([r[1] for r in c.ranges], dtype=np.float64)
    if not numpy_range:
        lo, hi = lo.astype(np.float32).astype(np.float64), hi.astype(np.float32).astype(np.float64)
    inside = ((p >= lo) & (p <= hi)).all(axis=1)
    assert np.array_equal(c.voxels.get_valid_values(pts.cuda()).cpu().numpy().reshape(-1), inside)
    coords, centres = pv.get_coordinates_and_points_in_grid(0.05, c.ranges)
    ref_val, ref_grad = gt(centres)
    cells = c._packed.cpu().reshape(21, 21, 2, 4)
    assert torch.equal(cells[:, :, 0], cells[:, :, 1]) and not cells[..., 3].any()      # two identical layers, no z slope
    cell_val, cell_grad = cells[:, :, 0, 0].reshape(-1), cells[:, :, 0, 1:3].reshape(-1, 2)
    # the cache evaluates the ground truth on the device (sdf.py:510): same cells up to the device's norm rounding
    assert torch.allclose(cell_val, ref_val, atol=1e-6) and torch.allclose(cell_grad, ref_grad, atol=1e-5)
    key = c.voxels.ensure_index_key(pts.cuda()).cpu().reshape(-1, 2)
    flat = c.voxels.ravel_multi_index(key, c.voxels.shape)
    # away from the half-cell planes the index is unambiguous: nearest cell centre
    res = (hi - lo) / 20
    t = (p - lo) / res
    clear = inside & (np.abs(t - np.floor(t) - 0.5) > 1e-3).all(axis=1)
    assert np.array_equal(key.numpy()[clear], np.rint(t[clear]).astype(np.int64))
    v, gr = val.cpu().reshape(-1), grad.cpu().reshape(-1, 2)
    assert torch.equal(v[clear], cell_val[flat[clear]]) and torch.equal(gr[clear], cell_grad[flat[clear]])
    assert torch.equal(c.voxels.raw_data.cpu(), cell_val)
    # out of range: distance to the (unpadded) bounding box of the circle, unit gradient pointing away from it
    bb = gt.surface_bounding_box().double().numpy()
    d = np.maximum(bb[:, 0] - p, 0) + np.maximum(p - bb[:, 1], 0)
    oob = ~inside
    assert np.allclose(v.numpy()[oob], np.linalg.norm(d[oob], axis=1), atol=1e-6)
    assert np.allclose(np.linalg.norm(gr.numpy()[oob], axis=1), 1.0, atol=1e-5)
    assert torch.equal(c.outside_surface(pts.cuda(), 0.0).cpu().reshape(-1)[clear], (cell_val[flat[clear]] > 0))
    with pytest.raises(ValueError):
        c(torch.zeros(4, 3).cuda())


def test_planar_cache_with_a_cell_count_that_three_does_not_divide_and_its_pickle(tmp_path):
    """20 x 20 = 400 cells: the gradient of a planar ground truth is (N, 2) and is stored and pickled as such (the
    reference: sdf.py:505 `sdf_grad.squeeze(0)`); round 2 reshaped it to (-1, 3), which 800 numbers do not allow.  The
    pickle is read back without querying the ground truth (sdf.py:484-500) and answers the same."""
    gt = CircleSDF(0.3)
    path = str(tmp_path / "planar_cache.pkl")
    rng = [(-0.5, 0.45), (-0.4, 0.55)]
    c = pv.CachedSDF("circle", 0.05, rng, gt, device="cuda", cache_path=path)
    assert c.voxels.shape == (20, 20)
    data = torch.load(path, weights_only=False)
    val, grad = data[c.name]
    assert tuple(val.shape) == (20, 20) and tuple(grad.shape) == (400, 2)
    pts = (torch.rand(4000, 2, generator=torch.Generator().manual_seed(3)) * 1.4 - 0.7).cuda()
    class BoxOnly(pv.ObjectFrameSDF):  # the cache must make querying the ground truth unnecessary
        def __call__(self, p):
            raise AssertionError("served from the pickle")

        def surface_bounding_box(self, **kw):
            return gt.surface_bounding_box(**kw)

    again = pv.CachedSDF("circle", 0.05, rng, BoxOnly(), device="cuda", cache_path=path)
    v1, g1 = c(pts)
    assert torch.equal(again._packed, c._packed)
    inside = c.voxels.get_valid_values(pts)
    v2, g2 = again(pts)
    assert torch.equal(v1, v2) and torch.equal(g1.nan_to_num(7.0), g2.nan_to_num(7.0)) and g1.shape == (4000, 2)
    assert inside.any() and (~inside).any()


@pytest.mark.parametrize("f64", [True, False])
def test_streaming_launch_equals_the_chunked_calls_and_the_oracle(f64):
    """More than 8M points take the STREAMING instantiation of the wave-tile kernel (non-temporal loads, one tile per wave on an
    uncapped grid, and -- since round 5 -- the round 1-4 look-up statements, where smaller launches use the composed kernels'
    shorter ones: csrc/grid_lookup.h cached_lookup<F64, STREAMING>).  Both must return the same bits for the same points: the
    big call against the same points in chunks below the threshold, and against the oracle on three runs of 20,000 points."""
    c = make_cached(f64=f64)
    P = (8 << 20) + 12_345  # a ragged end on top
    lo = np.array([r[0] for r in c.ranges]) - 0.05
    hi = np.array([r[1] for r in c.ranges]) + 0.05
    import workloads
    pts = workloads.uniform_points_device(P, lo, hi, seed=11)
    pts[123] = float("nan")
    pts[P - 7, 1] = float("inf")
    val, grad = c(pts)  # one streaming launch
    assert val.shape == (P,) and grad.shape == (P, 3)
    step = 3_000_000
    for a in range(0, P, step):  # the same points through the non-streaming instantiation
        v, g = c(pts[a:a + step].contiguous())
        assert torch.equal(v.nan_to_num(7.0), val[a:a + step].nan_to_num(7.0))
        assert torch.equal(g.nan_to_num(7.0), grad[a:a + step].nan_to_num(7.0))
    og = H.oracle_grid_from_cached(c)
    for a in (0, P // 2, P - 20_000):
        ov, ogr, _ = oracle.cached_query(og, pts[a:a + 20_000].cpu().numpy())
        assert np.array_equal(val[a:a + 20_000].cpu().numpy(), ov, equal_nan=True)
        assert np.array_equal(grad[a:a + 20_000].cpu().numpy(), ogr, equal_nan=True)
    # ... and the instantiation that also writes the out-of-range mask (what LOOKUP_GT_SDF asks for), through the C entry
    import ctypes
    lib = pv._lib.load()
    v2, g2 = torch.empty_like(val), torch.empty_like(grad)
    oob = torch.empty((P,), dtype=torch.uint8, device="cuda")
    desc = c._grid_desc()
    pv._lib.check(lib.pvamd_cached_query(ctypes.byref(desc), pv._lib.ptr(pts), P, pv._lib.ptr(v2), pv._lib.ptr(g2), pv._lib.ptr(oob),
                                         pv._lib.stream_ptr()), "pvamd_cached_query")
    assert torch.equal(v2.nan_to_num(7.0), val.nan_to_num(7.0)) and torch.equal(g2.nan_to_num(7.0), grad.nan_to_num(7.0))
    assert torch.equal(oob.bool(), ~c.voxels.get_valid_values(pts))


def test_query_into_follows_a_reassigned_bounding_box_and_a_repacked_cache():
    """query_into kept the first descriptor it saw; `c.bb = ...` (and a re-packed cache) changed what
    __call__ answered but not what query_into did.  Both now read the descriptor cache that __setattr__ invalidates."""
    c = make_cached()
    pts = query_points(c, 20_000, seed=3).cuda()
    val, grad = torch.empty(20_000, device="cuda"), torch.empty(20_000, 3, device="cuda")
    c.query_into(pts, val, grad)
    v0, g0 = c(pts)
    assert torch.equal(val, v0) and torch.equal(grad.view(torch.int32), g0.view(torch.int32))
    c.bb = c.bb + 0.01
    c.query_into(pts, val, grad)
    v1, g1 = c(pts)
    assert not torch.equal(v1, v0)  # the out-of-range half of the points sees the moved box
    assert torch.equal(val, v1) and torch.equal(grad.view(torch.int32), g1.view(torch.int32))
    oval, ograd, _ = oracle.cached_query(H.oracle_grid_from_cached(c), pts.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True) and np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    c._packed = c._packed.clone() * 2.0  # a re-packed cache at another address
    c.query_into(pts, val, grad)
    v2, g2 = c(pts)
    assert torch.equal(val, v2) and torch.equal(grad.view(torch.int32), g2.view(torch.int32)) and not torch.equal(v2, v1)


@pytest.mark.parametrize("oob", [pv.OutOfBoundsStrategy.BOUNDING_BOX, pv.OutOfBoundsStrategy.LOOKUP_GT_SDF])
@pytest.mark.parametrize("f64", [True, False])
def test_nan_and_inf_points_through_the_wave_tile_kernel_in_both_oob_modes(oob, f64):
    """the med3 range test and the NaN-ignoring max of the index-estimate check (grid_lookup.h cached_lookup, the
    statements of launches up to 8M points) had NaN / +-inf coverage only through the composed kernels.  40,000 points take the
    wave-tile kernel; non-finite coordinates sit in every lane position of a tile, alone and together."""
    c = make_cached(f64=f64, oob=oob)
    n = 40_000
    pts = query_points(c, n, seed=21)
    bad = [float("nan"), float("inf"), -float("inf")]
    for i in range(0, 1024):
        pts[7 * i % n, i % 3] = bad[i % 3]
    pts[20_001] = float("nan")
    pts[20_002] = float("inf")
    pts[20_003, 0], pts[20_003, 1] = float("nan"), -float("inf")
    val, grad = c(pts.cuda())
    og = H.oracle_grid_from_cached(c)
    oval, ograd, ooob = oracle.cached_query(og, pts.numpy())
    assert ooob[20_001] and ooob[20_002] and ooob[20_003]
    if oob == pv.OutOfBoundsStrategy.LOOKUP_GT_SDF:  # sdf.py:552-554: the ground truth answers the out-of-range subset
        idx = np.nonzero(ooob)[0]
        v_gt, g_gt = c.gt_sdf(pts[idx].cuda())  # on the device, as CachedSDF.__call__ asks it
        oval[idx], ograd[idx] = v_gt.cpu().numpy(), g_gt.cpu().numpy()
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def _kernel_boundaries():
    """Point counts either side of every threshold of pvamd_cached_query's kernel choice (found by scanning
    pvamd_cached_query_kernel in steps of 4096 and bisecting inside a step, so the test follows the thresholds if they are
    retuned; a kernel may serve more than one range), plus a ragged size after each threshold."""
    lib = pv._lib.load()
    kind = lambda p: int(lib.pvamd_cached_query_kernel(p))
    top = (8 << 20) + 4096
    sizes, prev = {1, top}, 1
    for p in range(4096, top + 1, 4096):
        if kind(p) != kind(prev):
            a, b = prev, p
            while b - a > 1:  # kind(a) == kind(prev) != kind(b)
                m = (a + b) // 2
                a, b = (m, b) if kind(m) == kind(prev) else (a, m)
            sizes.update({a, b, min(b + 191, top)})
        prev = p
    return sorted(sizes)


@pytest.mark.parametrize("f64", [True, False])
def test_every_kernel_of_the_size_dispatch_matches_the_oracle_bitwise(f64):
    """Round 6: pvamd_cached_query picks one of seven kernels / instantiations by the point count (csrc/cached.hip cq_kind).  Every
    one of them, at the first and last size it serves and at a ragged size, on buffers at odd dword offsets, with and without the
    out-of-range mask, against the oracle: same bits."""
    import ctypes
    c = make_cached(f64=f64)
    og = H.oracle_grid_from_cached(c)
    lib = pv._lib.load()
    sizes = _kernel_boundaries()
    seen = {int(lib.pvamd_cached_query_kernel(p)) for p in sizes}
    assert seen == set(range(7)), seen  # PVAMD_CQ_KERNEL_*: all seven
    top = max(sizes)
    lo = np.array([r[0] for r in c.ranges]) - 0.05
    hi = np.array([r[1] for r in c.ranges]) + 0.05
    import workloads
    pool = workloads.uniform_points_device(top + 1, lo, hi, seed=31)  # one pool; every size reads a window at a 12-byte offset
    pool[5, 0] = float("nan")
    pool[77, 2] = float("inf")
    opts = pool.cpu().numpy()
    oval, ograd, ooob = oracle.cached_query(og, opts)
    vbuf = torch.empty(top + 3, device="cuda")
    gbuf = torch.empty(3 * top + 5, device="cuda")
    obuf = torch.empty(top + 1, dtype=torch.uint8, device="cuda")
    desc = c._grid_desc()
    for i, P in enumerate(sizes):
        first = 1 + (i % 2)  # windows starting at point 1 or 2: 12- and 24-byte offsets
        if first + P > top + 1:
            first = top + 1 - P
        pts = pool[first:first + P]
        val, grad = vbuf[1:1 + P], gbuf[1:1 + 3 * P]  # odd dword addresses
        vbuf.fill_(-7.0); gbuf.fill_(-7.0); obuf.fill_(9)
        want_oob = i % 3 != 0
        pv._lib.check(lib.pvamd_cached_query(ctypes.byref(desc), pv._lib.ptr(pts), P, pv._lib.ptr(val), pv._lib.ptr(grad),
                                             pv._lib.ptr(obuf) if want_oob else None, pv._lib.stream_ptr()), "pvamd_cached_query")
        torch.cuda.synchronize()
        assert np.array_equal(val.cpu().numpy(), oval[first:first + P], equal_nan=True), P
        assert np.array_equal(grad.cpu().numpy().reshape(P, 3), ograd[first:first + P], equal_nan=True), P
        assert float(vbuf[0]) == -7.0 and float(vbuf[1 + P]) == -7.0 and float(gbuf[0]) == -7.0 and float(gbuf[1 + 3 * P]) == -7.0, P
        if want_oob:
            assert np.array_equal(obuf[:P].cpu().numpy().astype(bool), ooob[first:first + P].astype(bool)), P
            assert int(obuf[P]) == 9  # nothing written past the end
