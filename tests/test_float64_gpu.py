"""-m gpu: float64 query points are looked up in float64.

Reference behaviour (sdf.py:535-571): the output dtype is the query dtype (:545-547); `(points - min) / resolution` and the
range test promote to float64 whatever dtype the range had (:537,540); the BOUNDING_BOX branch runs on
self.bb.to(float64) (:556-571).  The HIP path: pvamd_cached_query_f64 / _outside_f64 / pvamd_voxel_index_f64; the oracle
twins: oracle.cached_query_f64 & co.  Everything bit-exact (float64 equality), including at half-voxel planes and range
edges that only float64 can resolve.
"""
import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from tests import helpers as H
from tests.test_cached_gpu import make_cached, query_points

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("f64_range", [True, False])
@pytest.mark.parametrize("n", [0, 1, 5, 1023, 200_003])
def test_float64_query_matches_oracle_bitwise(f64_range, n):
    c = make_cached(f64=f64_range)
    og = H.oracle_grid_from_cached(c)
    pts = query_points(c, n, seed=n + 3).double() + 1e-9  # not representable in float32
    val, grad = c(pts.cuda())
    assert val.dtype == torch.float64 and grad.dtype == torch.float64 and val.shape == (n,) and grad.shape == (n, 3)
    oval, ograd, ooob = oracle.cached_query_f64(og, pts.numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    if n > 1000:
        assert 0.05 < ooob.mean() < 0.95  # both branches


@pytest.mark.parametrize("f64_range", [True, False])
def test_half_voxel_planes_only_float64_resolves(f64_range):
    """Points 1e-12 m either side of every half-voxel plane: as float32 the two collapse onto one value (one index), in
    float64 they are different voxels -- and the product must say so, bit for bit with the oracle."""
    c = make_cached(f64=f64_range)
    og = H.oracle_grid_from_cached(c)
    v = c._view
    mn, res = v.dmin.numpy(), v.dres.numpy()
    shape = np.array(v.shape)
    rng = np.random.default_rng(0)
    n = 20_000
    k = rng.integers(0, shape - 1, size=(n, 3))
    centre = mn + (k + rng.random((n, 3)) * 0.6 + 0.2) * res  # somewhere inside cell k..k+1, away from the planes
    d = rng.integers(0, 3, size=n)
    plane = mn[d] + (k[np.arange(n), d] + 0.5) * res[d]
    below, above = centre.copy(), centre.copy()
    below[np.arange(n), d] = plane - 1e-12
    above[np.arange(n), d] = plane + 1e-12
    pts = np.concatenate((below, above))
    key = c.voxels.ensure_index_key(torch.from_numpy(pts).cuda()).cpu().numpy()
    okey, oflat, ovalid = oracle.voxel_index_f64(og, pts)
    assert np.array_equal(key, okey)
    assert ovalid.all()
    kd_below, kd_above = key[:n][np.arange(n), d], key[n:][np.arange(n), d]
    assert np.array_equal(kd_above, kd_below + 1)  # float64 separates the two sides of every plane ...
    key32 = c.voxels.ensure_index_key(torch.from_numpy(pts.astype(np.float32)).cuda()).cpu().numpy()
    # ... float32 cannot (2e-12 m apart: the two sides round to the same float32 except where the plane itself is a float32
    # rounding midpoint: fmin + (k + 0.5) * fres has about one bit more than a float32)
    assert (key32[:n] == key32[n:]).all(axis=1).mean() > 0.98
    val, grad = c(torch.from_numpy(pts).cuda())
    oval, ograd, _ = oracle.cached_query_f64(og, pts)
    assert np.array_equal(val.cpu().numpy(), oval) and np.array_equal(grad.cpu().numpy(), ograd)
    # what casting the query to float32 first (round 1's behaviour) would have cost: a wrong voxel for about half the points
    val32, _ = c(torch.from_numpy(pts.astype(np.float32)).cuda())
    wrong = (val32.double().cpu().numpy() != oval).mean()
    assert wrong > 0.2


@pytest.mark.parametrize("f64_range", [True, False])
def test_range_edges_and_outside_surface_in_float64(f64_range):
    c = make_cached(f64=f64_range)
    og = H.oracle_grid_from_cached(c)
    v = c._view
    mn, mx = v.dmin.numpy(), v.dmax.numpy()
    mid = 0.5 * (mn + mx)
    rows = []
    for d in range(3):
        for edge in (mn[d], mx[d]):
            for delta in (0.0, -1e-13, 1e-13, -1e-9, 1e-9):
                p = mid.copy()
                p[d] = edge + delta
                rows.append(p)
    pts = np.array(rows)
    valid = c.voxels.get_valid_values(torch.from_numpy(pts).cuda()).cpu().numpy()
    _, _, ovalid = oracle.voxel_index_f64(og, pts)
    assert np.array_equal(valid, ovalid) and ovalid.any() and (~ovalid).any()
    val, grad = c(torch.from_numpy(pts).cuda())
    oval, ograd, _ = oracle.cached_query_f64(og, pts)
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    big = query_points(c, 50_000, seed=11).double() * (1 + 1e-12)
    out = c.outside_surface(big.cuda(), surface_level=0.003)
    assert np.array_equal(out.cpu().numpy(), oracle.cached_outside_f64(og, big.numpy(), 0.003))


def test_bounding_box_branch_uses_the_float64_box():
    """sdf.py:556-557: bb is cast to the query dtype, so float64 queries measure against the un-rounded box."""
    c = make_cached(f64=True)
    far = torch.tensor([[1.0 + 1e-10, -2.0, 0.5], [0.3, 0.3 + 1e-11, 4.0]], dtype=torch.float64)
    val, grad = c(far.cuda())
    bb = c.bb.double().cpu().numpy()
    p = far.numpy()
    lo, hi = np.maximum(bb[:, 0] - p, 0), np.maximum(p - bb[:, 1], 0)
    ref = np.linalg.norm(lo + hi, axis=-1)
    assert np.allclose(val.cpu().numpy(), ref, rtol=0, atol=1e-15)
    assert np.allclose(np.linalg.norm(grad.cpu().numpy(), axis=-1), 1.0, atol=1e-15)


def test_lookup_gt_strategy_keeps_the_query_dtype():
    obj = pv.MeshObjectFactory(H.mesh_path("probe.obj"))
    c = pv.CachedSDF("probe", 0.02, obj.bounding_box(padding=0.05), pv.MeshSDF(obj),
                     out_of_bounds_strategy=pv.OutOfBoundsStrategy.LOOKUP_GT_SDF, device="cuda", cache_path=None)
    lo = np.array([r[0] for r in c.ranges]) - 0.1
    hi = np.array([r[1] for r in c.ranges]) + 0.1
    pts = H.uniform_points(5000, lo, hi, seed=5).double()
    val, grad = c(pts.cuda())
    assert val.dtype == torch.float64 and grad.dtype == torch.float64
    v32, g32 = c(pts.float().cuda())
    inb = c.voxels.get_valid_values(pts.cuda())
    # out of range: ground truth from the mesh kernel (float32 arithmetic, sdf.py:132) widened to the query dtype
    assert torch.equal(val[~inb], v32[~inb].double()) and (~inb).any()


def test_outside_surface_compares_in_float32_whatever_the_query_dtype():
    """sdf.py:601 `raw_data[flat] > surface_level`: the cache is a float32 tensor and a python scalar does not promote it,
    so a level that float32 cannot represent (0.1) is rounded first -- a voxel holding float32(0.1) is NOT above level 0.1,
    for float32 and float64 query points alike (round 2 compared in float64 for float64 points)."""

    class Constant(pv.ObjectFrameSDF):
        def __call__(self, p):
            return torch.full(p.shape[:-1], 0.1, dtype=p.dtype, device=p.device), torch.zeros_like(p)

        def surface_bounding_box(self, **kw):
            return torch.tensor([[-0.1, 0.1]] * 3, dtype=torch.float64)

    c = pv.CachedSDF("const", 0.05, np.array([[-0.5, 0.5]] * 3), Constant(), device="cuda", cache_path=None)
    assert float(c.voxels.raw_data[0]) == float(np.float32(0.1)) > 0.1
    pts = (torch.rand(1000, 3, dtype=torch.float64) - 0.5) * 0.9
    for q in (pts, pts.float()):
        out = c.outside_surface(q.cuda(), surface_level=0.1)
        assert not out.any()
        assert c.outside_surface(q.cuda(), surface_level=0.0999999).all()
    og = H.oracle_grid_from_cached(c)
    assert not oracle.cached_outside_f64(og, pts.numpy(), 0.1).any() and not oracle.cached_outside(og, pts.float().numpy(), 0.1).any()


@pytest.mark.parametrize("A", [1, 7])
def test_composed_and_robot_keep_float64_points_in_float64(A):
    """sdf.py:395-431 over sdf.py:545-547: a composition answers float64 points in float64 -- the transform, every leaf's
    index arithmetic / range test / bounding-box branch and the gradient rotation (round 2 computed in float32 and cast
    back, which picks other voxels near the half-voxel planes).  Bit-exact vs the oracle's float64 composition."""
    S = 5
    leaves = [make_cached(f64=(s % 2 == 0)) for s in range(S)]
    tfm = H.random_rigid(S * A, seed=31 + A, trans=0.2).double()
    tfm = tfm + 1e-11 * torch.randn(tfm.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1)) * \
        torch.tensor([[1.0], [1.0], [1.0], [0.0]], dtype=torch.float64)  # digits only float64 carries; last row stays 0 0 0 1
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,) if A > 1 else None, known_rigid=True)
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(30_000, 3, dtype=torch.float64, generator=g) - 0.5) * 0.9
    val, grad = comp(pts.cuda())
    assert val.dtype == torch.float64 and grad.dtype == torch.float64
    assert val.shape == ((A, 30_000) if A > 1 else (30_000,))
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    oval, ograd, oleaf = oracle.composed_query_f64(ogrids, tfm.numpy(), A, pts.numpy())
    assert np.array_equal(val.cpu().numpy().reshape(A, -1), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy().reshape(A, -1, 3), ograd, equal_nan=True)
    assert len(np.unique(oleaf)) == S
    # and it is NOT what the float32 path gives
    v32, _ = comp(pts.float().cuda())
    assert v32.dtype == torch.float32
    assert (v32.double().cpu().numpy().reshape(A, -1) != oval).mean() > 0.01


@pytest.mark.parametrize("A", [1, 3])
def test_float64_points_over_mesh_leaves_are_transformed_in_float64(A):
    """sdf.py:395-431 with MeshSDF leaves (the reference's own tests/test_sdf.py:61-80 composition) and float64 points: the
    transform runs in float64 and its RESULT is what the mesh query rounds to float32 (sdf.py:132) -- one rounding, where a
    float32 transform rounds every product -- the gradient comes back through R^T in float64 and the first minimum wins.
    Restated here in numpy around the oracle's mesh query, operation for operation: equal bit for bit."""
    from pytorch_volumetric_amd import mesh_io
    objs = [pv.MeshObjectFactory(H.mesh_path("box_template.obj"), scale=0.2), pv.MeshObjectFactory(H.mesh_path("probe.obj"))]
    objs[0].jitter_seed, objs[1].jitter_seed = 5, 9
    leaves = [pv.MeshSDF(o) for o in objs]
    S, P = 2, 3000
    tfm = H.random_rigid(S * A, seed=7 + A, trans=0.05).double()
    tfm = tfm + 1e-11 * torch.randn(tfm.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(2)) * \
        torch.tensor([[1.0], [1.0], [1.0], [0.0]], dtype=torch.float64)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,) if A > 1 else None, known_rigid=True)
    pts = (torch.rand(P, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(4)) - 0.5) * 0.5
    val, grad = comp(pts.cuda())
    assert val.dtype == torch.float64 and grad.dtype == torch.float64 and val.shape == ((A, P) if A > 1 else (P,))
    p, m = pts.numpy(), tfm.numpy().reshape(S, A, 4, 4)
    best_v = best_g = None
    rounded_differently = 0
    for i in range(S):
        M = m[i]
        x = np.stack([M[:, r, 0, None] * p[None, :, 0] + M[:, r, 1, None] * p[None, :, 1] + M[:, r, 2, None] * p[None, :, 2]
                      + M[:, r, 3, None] for r in range(3)], axis=-1)
        x32 = x.astype(np.float32).reshape(-1, 3)
        m32, p32 = M.astype(np.float32), p.astype(np.float32)
        in_f32 = np.stack([m32[:, r, 0, None] * p32[None, :, 0] + m32[:, r, 1, None] * p32[None, :, 1]
                           + m32[:, r, 2, None] * p32[None, :, 2] + m32[:, r, 3, None] for r in range(3)], axis=-1)
        rounded_differently += int((in_f32.reshape(-1, 3) != x32).sum())
        _, od, og, _, _ = oracle.mesh_query(H.oracle_mesh_from_factory(objs[i]), x32, seed=objs[i].jitter_seed)
        v, g = od.astype(np.float64).reshape(A, P), og.astype(np.float64).reshape(A, P, 3)
        g = np.stack([M[:, 0, j, None] * g[..., 0] + M[:, 1, j, None] * g[..., 1] + M[:, 2, j, None] * g[..., 2]
                      for j in range(3)], axis=-1)
        if best_v is None:
            best_v, best_g = v, g
        else:
            take = (v < best_v) | (np.isnan(v) & ~np.isnan(best_v))
            best_v, best_g = np.where(take, v, best_v), np.where(take[..., None], g, best_g)
    assert rounded_differently > 0  # the float32 transform is a different set of query points
    assert np.array_equal(val.cpu().numpy().reshape(A, P), best_v, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy().reshape(A, P, 3), best_g, equal_nan=True)
    v32, g32 = comp(pts.float().cuda())  # float32 points: the float32 kernels, float32 out
    assert v32.dtype == torch.float32 and np.allclose(v32.double().cpu().numpy().reshape(A, P), best_v, atol=1e-5)
