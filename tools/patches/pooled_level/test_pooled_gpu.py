"""-m gpu: the pooled level of a large leaf grid (pvamd_pool_grid: (min, max) of the values over 4 x 4 x 4 voxel blocks dilated by one
voxel) and the per-lane composed kernel's two-pass walk over it -- a tuning aid: with and without it every result is the same, bit
for bit (sdf.py:414-421: a leaf whose values around the point are all above another leaf's cannot be the argmin)."""
import ctypes

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _numpy_pool(vals):
    nx, ny, nz = vals.shape
    nb = [(n + 3) // 4 for n in vals.shape]
    out = np.empty((*nb, 2), dtype=np.float32)
    for bx in range(nb[0]):
        for by in range(nb[1]):
            for bz in range(nb[2]):
                blk = vals[max(4 * bx - 1, 0):min(4 * bx + 4, nx - 1) + 1, max(4 * by - 1, 0):min(4 * by + 4, ny - 1) + 1,
                           max(4 * bz - 1, 0):min(4 * bz + 4, nz - 1) + 1]
                out[bx, by, bz] = (-np.inf, np.inf) if np.isnan(blk).any() else (blk.min(), blk.max())
    return out.reshape(-1, 2)


def test_pool_grid_is_the_dilated_block_min_max():
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for shape in ((9, 6, 13), (4, 4, 4), (2, 3, 5), (17, 8, 1)):
        vals = rng.normal(size=shape).astype(np.float32)
        vals[rng.random(shape) < 0.01] = np.nan
        vals[rng.random(shape) < 0.01] = np.inf
        packed = torch.zeros((vals.size, 4), dtype=torch.float32, device="cuda")
        packed[:, 0] = torch.from_numpy(vals.reshape(-1)).cuda()
        nb = int(np.prod([(n + 3) // 4 for n in shape]))
        pool = torch.full((nb, 2), 7.0, dtype=torch.float32, device="cuda")
        _lib.check(lib.pvamd_pool_grid(_lib.ptr(packed), *shape, _lib.ptr(pool), _lib.stream_ptr()), "pvamd_pool_grid")
        assert np.array_equal(pool.cpu().numpy(), _numpy_pool(vals)), shape


def test_attach_pool_wants_a_finalized_descriptor_and_a_512_byte_multiple():
    lib = _lib.load()
    gt = H.drill_like_gt()
    c = pv.CachedSDF("leaf", 0.02, H.padded_range(H.DRILL_BB, 0.1), gt, device="cuda", cache_path=None)
    d = _lib.GridDesc.from_buffer_copy(c._grid_desc())
    base = d.vox
    assert lib.pvamd_grid_attach_pool(ctypes.byref(d), ctypes.c_void_p(base + 512 * 5)) == 0 and d.pool_off512 == 5
    assert lib.pvamd_grid_attach_pool(ctypes.byref(d), ctypes.c_void_p(base - 512 * 3)) == 0 and d.pool_off512 == -3
    assert lib.pvamd_grid_attach_pool(ctypes.byref(d), ctypes.c_void_p(base + 100)) < 0  # PVAMD_E_ALIGN
    assert lib.pvamd_grid_attach_pool(ctypes.byref(d), None) == 0 and d.pool_off512 == 0
    raw = _lib.GridDesc()
    assert lib.pvamd_grid_attach_pool(ctypes.byref(raw), ctypes.c_void_p(512)) != 0  # not finalized


def _big_leaves(n, res=0.01, pad=0.45, seed=0):
    """leaf caches above the 1 MB / 4 MB thresholds (a drill-sized box with wide padding: ~2-3 M voxels would be slow to build from
    a mesh; an analytic ground truth fills them)"""
    gt = H.drill_like_gt()
    return [pv.CachedSDF(f"big{seed}_{s}", res, H.padded_range(H.DRILL_BB, pad), gt, device="cuda", cache_path=None) for s in range(n)]


def _both_ways(comp, pts):
    comp.pooled_leaves = True
    a = comp(pts)
    grids_pooled = comp._leaf_grids(pts.device)
    descs = np.frombuffer(grids_pooled.cpu().numpy().tobytes(), dtype=np.uint8)
    comp.pooled_leaves = False
    b = comp(pts)
    comp.pooled_leaves = True
    return a, b, descs


def test_with_and_without_pooled_levels_the_per_lane_kernel_returns_the_same_bits():
    leaves = _big_leaves(3)
    assert all(l._packed.numel() * 4 > (1 << 20) for l in leaves)
    # make the leaves differ (the same analytic field in three copies would tie everywhere): scale / offset the values
    leaves[1]._packed[:, 0] += 0.013
    leaves[2]._packed[:, 0] *= 1.1
    leaves[2]._packed[::9973, 0] = float("nan")  # NaN records win the argmin (sdf.py:421): their blocks are never skipped
    lo, hi = [r[0] - 0.05 for r in leaves[0].ranges], [r[1] + 0.05 for r in leaves[0].ranges]
    pts = H.uniform_points(20_011, lo, hi, seed=2).cuda()  # < 32,768 points: the per-lane kernel; some points out of range
    pts[5] = float("nan")
    pts[6, 2] = float("inf")
    for batch in (None, (5,)):
        A = 1 if batch is None else 5
        comp = pv.ComposedSDF(leaves, None)
        comp.set_transforms(pv.Transform3d(matrix=H.random_rigid(3 * A, seed=11 + A, trans=0.15)), batch_dim=batch)
        (v1, g1), (v2, g2), _ = _both_ways(comp, pts)
        assert all(l.__dict__.get("_pool") is not None for l in leaves)
        assert np.array_equal(v1.cpu().numpy(), v2.cpu().numpy(), equal_nan=True)
        assert np.array_equal(g1.cpu().numpy(), g2.cpu().numpy(), equal_nan=True)
        # an ordered slice as well (whole waves inside every leaf's range)
        xs = torch.linspace(float(lo[0]) + 0.1, float(hi[0]) - 0.1, 150)
        zs = torch.linspace(float(lo[2]) + 0.1, float(hi[2]) - 0.1, 100)
        ys = torch.tensor([0.5 * float(lo[1] + hi[1])], dtype=torch.float32)
        sl = torch.stack(torch.meshgrid(xs, ys, zs, indexing="ij"), dim=-1).reshape(-1, 3).contiguous().cuda()
        (v1, g1), (v2, g2), _ = _both_ways(comp, sl)
        assert np.array_equal(v1.cpu().numpy(), v2.cpu().numpy(), equal_nan=True)
        assert np.array_equal(g1.cpu().numpy(), g2.cpu().numpy(), equal_nan=True)
    # ... and against the oracle on a sample
    from oracle import oracle
    comp = pv.ComposedSDF(leaves, None)
    tfm = H.random_rigid(3 * 2, seed=5, trans=0.15)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(2,))
    sample = pts[:3000].contiguous()
    v, g = comp(sample)
    ov, og, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), 2, sample.cpu().numpy())
    assert np.array_equal(v.cpu().numpy(), ov, equal_nan=True) and np.array_equal(g.cpu().numpy(), og, equal_nan=True)


def test_a_write_into_the_cache_re_derives_the_pooled_level():
    leaves = _big_leaves(2, seed=1)
    leaves[1]._packed[:, 0] += 0.02
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=H.random_rigid(2, seed=3, trans=0.1)), batch_dim=None)
    lo, hi = [r[0] + 0.05 for r in leaves[0].ranges], [r[1] - 0.05 for r in leaves[0].ranges]
    pts = H.uniform_points(9000, lo, hi, seed=4).cuda()
    v0, _ = comp(pts)
    pool0 = leaves[1].__dict__["_pool"]
    # in place, through the public view of the reference's API: leaf 1 now lies far BELOW leaf 0 everywhere
    leaves[1].voxels.raw_data[:] = leaves[1].voxels.raw_data - 5.0
    v1, g1 = comp(pts)
    assert leaves[1].__dict__["_pool"] is not pool0
    comp.pooled_leaves = False
    v2, g2 = comp(pts)
    assert torch.equal(v1, v2) and torch.equal(g1, g2) and not torch.equal(v0, v1)
    assert float((v1 < -4.0).float().mean()) > 0.5  # (points that the transform takes out of leaf 1's range keep leaf 0)
