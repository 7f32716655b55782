"""-m gpu: randomized differential tests, HIP path vs CPU oracle, over many small random configurations -- grid
shapes and resolutions, ranges far from the origin (where the fp32 index estimate has to fall back to the exact
division often), float32/float64 index arithmetic, random meshes and poses, awkward point counts."""
import os

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from oracle import oracle
from pytorch_volumetric_amd import mesh_io
from tests import helpers as H

FUZZ_SCALE = int(os.environ.get("PVAMD_FUZZ_SCALE", "1"))  # PVAMD_FUZZ_SCALE=30 pytest ... for a long campaign
FUZZ_BASE = int(os.environ.get("PVAMD_FUZZ_BASE", "0"))    # first seed: a second campaign draws cases the first did not
pytestmark = pytest.mark.gpu


class BoxGT(pv.ObjectFrameSDF):
    """analytic box SDF as ground truth for arbitrary grids"""

    def __init__(self, lo, hi):
        self.lo, self.hi = torch.tensor(lo, dtype=torch.float64), torch.tensor(hi, dtype=torch.float64)

    def __call__(self, p):
        c, h = ((self.lo + self.hi) / 2).to(p.device), ((self.hi - self.lo) / 2).to(p.device)
        q = (p.double() - c).abs() - h
        v = q.clamp(min=0).norm(dim=-1) + q.max(dim=-1).values.clamp(max=0)
        g = torch.nn.functional.normalize(torch.sign(p.double() - c) * (q >= q.max(dim=-1, keepdim=True).values), dim=-1)
        return v.to(p.dtype), g.to(p.dtype)

    def surface_bounding_box(self, padding=0., padding_ratio=0.):
        return torch.stack((self.lo - padding, self.hi + padding), dim=1)


def random_cached(rng, f64, far=False):
    centre = rng.uniform(-1, 1, 3) * (1000.0 if far else 1.0)
    half = rng.uniform(0.05, 0.4, 3)
    res = float(rng.choice([0.013, 0.02, 0.05, 0.1]))
    pad = float(rng.uniform(0.0, 0.3))
    lo, hi = centre - half, centre + half
    rng_np = np.stack((lo - pad, hi + pad), axis=1)
    rng_in = rng_np if f64 else [(float(a), float(b)) for a, b in rng_np]
    c = pv.CachedSDF("fuzz", res, rng_in, BoxGT(lo, hi), device="cuda", cache_path=None)
    return c, rng_np


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + 12 * FUZZ_SCALE))
def test_cached_query_fuzz(seed):
    rng = np.random.default_rng(seed)
    f64, far = bool(seed % 2), seed % 3 == 0
    c, r = random_cached(rng, f64, far)
    og = H.oracle_grid_from_cached(c)
    n = int(rng.choice([1, 63, 255, 256, 257, 4097, 16_385, 50_001, 200_001]))
    if seed % 8 == 5:  # the larger kernels of pvamd_cached_query's size dispatch (csrc/cached.hip cq_kind), ragged sizes
        n = int(rng.choice([950_003, 1_048_576, 1_200_001, 2_000_003, 2_500_001]))
    span = r[:, 1] - r[:, 0]
    pts = (r[:, 0] - 0.2 * span + rng.random((n, 3)) * 1.4 * span).astype(np.float32)
    # sprinkle exact voxel centres, half-voxel planes, range corners and specials
    k = min(n, 40)
    view = c._view
    mn = (view.dmin if f64 else view.fmin).double().numpy()
    rs = (view.dres if f64 else view.fres).double().numpy()
    idx = rng.integers(0, np.array(view.shape), size=(k, 3))
    pts[:k] = (mn + (idx + rng.choice([0.0, 0.5, 0.4999999, 0.5000001], size=(k, 3))) * rs).astype(np.float32)
    if n > 45:
        pts[41] = [np.nan, pts[41, 1], pts[41, 2]]
        pts[42] = [np.inf, 0, 0]
        pts[43] = (r[:, 0]).astype(np.float32)
        pts[44] = (r[:, 1]).astype(np.float32)
    t = torch.from_numpy(pts).cuda()
    val, grad = c(t)
    oval, ograd, ooob = oracle.cached_query(og, pts)
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    key = c.voxels.ensure_index_key(t).cpu().numpy()
    okey, _, ovalid = oracle.voxel_index(og, pts)
    finite = np.isfinite(pts).all(axis=1)
    assert np.array_equal(key[finite], okey[finite])
    assert np.array_equal(c.voxels.get_valid_values(t).cpu().numpy(), ovalid)
    assert np.array_equal(c.outside_surface(t, 0.01).cpu().numpy(), oracle.cached_outside(og, pts, 0.01))


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + 8 * FUZZ_SCALE))
def test_composed_query_fuzz(seed):
    rng = np.random.default_rng(100 + seed)
    S, A = int(rng.integers(1, 12)), int(rng.choice([1, 2, 5]))
    leaves = [random_cached(rng, bool(rng.integers(0, 2)))[0] for _ in range(S)]
    tfm = H.random_rigid(S * A, seed=seed, trans=1.5)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(tfm, batch_dim=(A,) if A > 1 else None)
    n = int(rng.choice([7, 256, 1000, 4096, 10_003]))
    pts = (rng.random((n, 3)) * 6 - 3).astype(np.float32)
    if seed % 3 == 1:  # spatially coherent tiles: clusters of 256 points -> the per-tile leaf mask and its refinement fire
        centres = (rng.random((max(n // 256, 1), 3)) * 6 - 3).repeat(256, axis=0)[:n]
        pts[: len(centres)] = (centres + rng.normal(scale=0.05, size=centres.shape)).astype(np.float32)[: n]
    if n > 300 and seed % 5 == 0:
        pts[257] = [np.nan, 0.1, 0.2]
        pts[3] = [np.inf, 0.0, 0.0]
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts)
    dev = torch.device("cuda", torch.cuda.current_device())
    comp._leaf_grids(dev)
    # every kernel / index mode the dispatcher can pick (include/pvamd.h): per-lane, wave-tile x {queued leaf loop, round-3
    # leaf loop (16), inline exact}, and the bucketed path (sort, packed records, un-permute) -- all must give the oracle's bits
    for flags, bucket in ((2, False), (2 | 16, False), (4, False), (4 | 16, False), (4 | 1, False), (0, True), (1, True), (16, True)):
        comp._query_flags = flags
        comp.bucket_points = bucket
        val, grad = comp(torch.from_numpy(pts).cuda())
        assert np.array_equal(val.cpu().numpy().reshape(A, -1), oval, equal_nan=True), (flags, bucket)
        assert np.array_equal(grad.cpu().numpy().reshape(A, -1, 3), ograd, equal_nan=True), (flags, bucket)
    if n >= _lib.group_chunk_points():
        # round 6: the chunk-grouped kernels -- the sort inside the workgroup (FORCE_FUSED) and the pre-pass + grouped pair
        comp.bucket_points = False
        for flags, group in ((_lib.COMPOSED_FORCE_FUSED, "auto"), (0, True)):
            comp._query_flags = flags
            comp.group_points = group
            val, grad = comp(torch.from_numpy(pts).cuda())
            assert np.array_equal(val.cpu().numpy().reshape(A, -1), oval, equal_nan=True), (flags, group)
            assert np.array_equal(grad.cpu().numpy().reshape(A, -1, 3), ograd, equal_nan=True), (flags, group)


def random_mesh(rng):
    kind = rng.integers(0, 3)
    if kind == 0:
        m = mesh_io.uv_sphere_mesh(float(rng.uniform(0.05, 0.5)), int(rng.integers(5, 40)), int(rng.integers(3, 20)),
                                   scale=tuple(rng.uniform(0.3, 1.5, 3)), center=tuple(rng.uniform(-1, 1, 3)))
    elif kind == 1:
        m = mesh_io.box_mesh(tuple(rng.uniform(0.05, 1.0, 3)), tuple(rng.uniform(-2, 2, 3)))
    else:  # triangle soup with no structure at all (open, self-intersecting): parity must still hold
        v = rng.uniform(-1, 1, (int(rng.integers(4, 300)), 3))
        f = rng.integers(0, len(v), (int(rng.integers(1, 700)), 3))
        f = f[(f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])]
        m = mesh_io.TriMesh(v, f if len(f) else np.array([[0, 1, 2]]))
    return m


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + 10 * FUZZ_SCALE))
def test_mesh_query_and_chamfer_fuzz(seed):
    rng = np.random.default_rng(200 + seed)
    obj = pv.MeshObjectFactory(mesh=random_mesh(rng))
    om = H.oracle_mesh_from_factory(obj)
    bb = obj.bounding_box(padding_ratio=0.6)
    n = int(rng.choice([1, 64, 65, 700, 2048, 5001]))
    pts = (bb[:, 0] + rng.random((n, 3)) * (bb[:, 1] - bb[:, 0])).astype(np.float32)
    obj.jitter_seed = 1000 + seed
    res = obj.object_frame_closest_point(torch.from_numpy(pts).cuda(), compute_normal=True)
    oc, od, og, of, on = oracle.mesh_query(om, pts, seed=1000 + seed)
    assert np.array_equal(obj._last_face_ids.cpu().numpy(), of)
    assert np.array_equal(res.closest.cpu().numpy(), oc)
    assert np.array_equal(res.distance.cpu().numpy(), od)
    assert np.array_equal(res.gradient.cpu().numpy(), og, equal_nan=True)
    B = int(rng.integers(1, 6))
    W = H.random_rigid(B, seed=seed, trans=0.3)
    err = pv.batch_chamfer_dist(W, torch.from_numpy(pts), obj, scale=10.0)
    oerr = oracle.chamfer_mesh(om, W.numpy(), pts, scale=10.0) / n
    assert np.allclose(err.double().numpy(), oerr, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + 6 * FUZZ_SCALE))
def test_rules_and_float64_compositions_fuzz(seed):
    """Random compositions whose leaves were built under a random index rule (pvamd_grid_t.rule), queried with float32 points
    through every kernel variant and with float64 points through pvamd_composed_query_f64: the oracle's bits each time."""
    from pytorch_volumetric_amd import voxel
    rng = np.random.default_rng(300 + seed)
    rule = int(rng.choice([0, 1, 2, 4, 1 | 2, 1 | 4, 8, 1 | 8]))
    prev = voxel.INDEX_RULE
    voxel.INDEX_RULE = rule
    try:
        S, A = int(rng.integers(1, 7)), int(rng.choice([1, 3]))
        leaves = [random_cached(rng, bool(rng.integers(0, 2)))[0] for _ in range(S)]
        assert all(l._view.rule == rule for l in leaves)
        tfm = H.random_rigid(S * A, seed=seed, trans=1.0)
        comp = pv.ComposedSDF(leaves, None)
        comp.set_transforms(tfm, batch_dim=(A,) if A > 1 else None)
        n = int(rng.choice([9, 300, 2049, 4100]))
        pts = (rng.random((n, 3)) * 4 - 2).astype(np.float32)
        # leaf-frame half-voxel planes and range edges of leaf 0 under configuration 0, brought back to the object frame
        v0 = leaves[0]._view
        k = min(n, 64)
        idx = rng.integers(-1, np.array(v0.shape) + 1, size=(k, 3))
        leaf_pts = v0.dmin.numpy() + (idx + rng.choice([0.0, 0.5, -0.5], size=(k, 3))) * v0.dres.numpy()
        inv = np.linalg.inv(tfm[0].double().numpy())
        pts[:k] = (leaf_pts @ inv[:3, :3].T + inv[:3, 3]).astype(np.float32)
        ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
        oval, ograd, _ = oracle.composed_query(ogrids, tfm.numpy(), A, pts)
        comp._leaf_grids(torch.device("cuda", torch.cuda.current_device()))
        for flags, group in ((2, "auto"), (4, "auto"), (4 | 1, "auto")) + \
                (((_lib.COMPOSED_FORCE_FUSED, "auto"), (0, True)) if n >= _lib.group_chunk_points() else ()):
            comp._query_flags = flags
            comp.group_points = group
            val, grad = comp(torch.from_numpy(pts).cuda())
            assert np.array_equal(val.cpu().numpy().reshape(A, -1), oval, equal_nan=True), (rule, flags, group)
            assert np.array_equal(grad.cpu().numpy().reshape(A, -1, 3), ograd, equal_nan=True), (rule, flags, group)
        comp.group_points = "auto"
        p64 = pts.astype(np.float64) + rng.normal(scale=1e-10, size=pts.shape)
        v64, g64 = comp(torch.from_numpy(p64).cuda())
        ov, og, _ = oracle.composed_query_f64(ogrids, tfm.double().numpy(), A, p64)
        assert v64.dtype == torch.float64
        assert np.array_equal(v64.cpu().numpy().reshape(A, -1), ov, equal_nan=True), rule
        assert np.array_equal(g64.cpu().numpy().reshape(A, -1, 3), og, equal_nan=True), rule
    finally:
        voxel.INDEX_RULE = prev
