"""-m gpu: VoxelGrid / ExpandingVoxelGrid / voxel_down_sample on the HIP gather / scatter kernels (SURVEY 8(f) rank 4,
reference voxel.py:42-171) vs the oracle's restatement and vs the generic torch path of the same container."""
import ctypes

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from pytorch_volumetric_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def oracle_grid_of(vg):
    """the oracle's view of a VoxelGrid's index arithmetic (storage handled in numpy by the caller)"""
    view = vg.voxels
    shape = view.shape
    rmin = np.array([b[0] for b in view._ranges])
    rmax = np.array([b[1] for b in view._ranges])
    f64 = view._min.dtype == torch.float64
    dt = np.float64 if f64 else np.float32
    dummy = np.zeros(shape, np.float32)
    return oracle.Grid(dummy, np.zeros((dummy.size, 3), np.float32), rmin.astype(dt), rmax.astype(dt), np.zeros((3, 2)),
                       index_f64=f64)


@pytest.mark.parametrize("f64_range", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bool])
def test_scatter_then_gather_matches_oracle(f64_range, dtype):
    rng = np.random.default_rng(3)
    lo, hi = np.array([-0.31, 0.07, 1.2]), np.array([0.45, 0.62, 1.9])
    ranges = np.stack((lo, hi), axis=1) if f64_range else [(float(a), float(b)) for a, b in zip(lo, hi)]
    vg = pv.VoxelGrid(0.013, ranges, dtype=dtype, device="cuda")
    assert vg.voxels._device_path(torch.zeros(1, 3, device="cuda"))
    og = oracle_grid_of(vg)
    P = 200_000
    pts = (lo - 0.1 + rng.random((P, 3)) * (hi - lo + 0.2)).astype(np.float32)  # ~40 % out of range, many duplicates
    pts[:500] = pts[500:1000]                                                    # exact duplicates, different values
    store = np.zeros(vg.voxels.shape, np.float32 if dtype == torch.float32 else np.bool_)
    if dtype == torch.float32:
        vals = rng.normal(size=P).astype(np.float32)
        vg[torch.from_numpy(pts).cuda()] = torch.from_numpy(vals).cuda()
        oracle.voxel_scatter(og, store, pts, vals)
    else:
        vg[torch.from_numpy(pts).cuda()] = 1
        oracle.voxel_scatter(og, store, pts, True)
    assert np.array_equal(vg.get_voxel_values().cpu().numpy(), store), "scatter: last writer in input order must win"
    q = (lo - 0.2 + rng.random((50_000, 3)) * (hi - lo + 0.4)).astype(np.float32)
    got = vg[torch.from_numpy(q).cuda()].cpu().numpy()
    assert np.array_equal(got, oracle.voxel_gather(og, store, q, 0))
    # batch dimensions carry through
    assert vg[torch.from_numpy(q).cuda().reshape(50, 1000, 3)].shape == (50, 1000)


def test_device_path_equals_the_generic_torch_path():
    """same container, storage on the CPU -> generic torch path; must agree with the HIP path bit for bit (one writer
    per voxel: torch's own index assignment does not define which of several writers wins)"""
    rng = np.random.default_rng(5)
    ranges = [(-1.0, 1.0), (-0.5, 0.5), (0.0, 0.7)]
    a, b = pv.VoxelGrid(0.05, ranges, device="cuda"), pv.VoxelGrid(0.05, ranges, device="cpu")
    centres = b.get_voxel_center_points()
    pick = torch.from_numpy(rng.permutation(len(centres))[:3000])
    pts = centres[pick] + torch.from_numpy(rng.uniform(-0.02, 0.02, (3000, 3)).astype(np.float32))
    pts = torch.cat((pts, torch.from_numpy(rng.uniform(1.3, 2.0, (500, 3)).astype(np.float32))))  # out of range: ignored
    vals = torch.from_numpy(rng.normal(size=3500).astype(np.float32))
    a[pts.cuda()] = vals.cuda()
    b[pts] = vals
    assert torch.equal(a.get_voxel_values().cpu(), b.get_voxel_values())
    assert torch.equal(a[pts.cuda()].cpu(), b[pts])
    pa, va = a.get_known_pos_and_values()
    pb, vb = b.get_known_pos_and_values()
    assert torch.equal(pa.cpu(), pb) and torch.equal(va.cpu(), vb)


def test_expanding_grid_and_down_sample_on_device():
    rng = np.random.default_rng(1)
    g = pv.ExpandingVoxelGrid(0.1, [(0.0, 1.0)] * 3, device="cuda")
    inside = torch.tensor([[0.5, 0.5, 0.5], [0.21, 0.88, 0.07]], device="cuda")
    g[inside] = torch.tensor([2.0, 3.0], device="cuda")
    far = torch.tensor([[1.73, -0.42, 0.5]], device="cuda")
    g[far] = torch.tensor([5.0], device="cuda")          # grows the range, keeps what was known
    assert g[inside].tolist() == [2.0, 3.0] and g[far].tolist() == [5.0]
    assert g.range_per_dim[0][1] >= 1.73 and g.range_per_dim[1][0] <= -0.42
    # down-sampling: one centre per occupied cell, every point within half a cell diagonal of a centre (test_voxel_sdf.py:29)
    cloud = torch.from_numpy(rng.normal(size=(200_000, 3)).astype(np.float32) * np.array([0.3, 0.2, 0.1], np.float32)).cuda()
    res = 0.02
    centres = pv.voxel_down_sample(cloud, res)
    on_cpu = pv.voxel_down_sample(cloud.cpu(), res)
    assert torch.equal(centres.cpu(), on_cpu)
    assert len(centres) < len(cloud)
    d = torch.cdist(cloud[:2000], centres).min(dim=1).values
    assert d.max().item() < res * np.sqrt(3) / 2 * 1.01


def test_cabi_rejects_missing_scratch_and_bad_shapes():
    lib = _lib.load()
    vg = pv.VoxelGrid(0.1, [(0.0, 1.0)] * 3, device="cuda")
    desc = vg.voxels._grid_desc()
    pts = torch.zeros(4, 3, device="cuda")
    vals = torch.zeros(4, device="cuda")
    store = vg.get_voxel_values()
    rc = lib.pvamd_voxel_scatter_f32(ctypes.byref(desc), _lib.ptr(store), _lib.ptr(pts), _lib.ptr(vals), 0.0, 4, None,
                                     _lib.stream_ptr())
    assert rc == -1  # per-point values need the owner scratch
    assert lib.pvamd_voxel_gather_f32(ctypes.byref(desc), _lib.ptr(store), _lib.ptr(pts), -1, 0.0, _lib.ptr(vals),
                                      _lib.stream_ptr()) == -2
    assert lib.pvamd_voxel_gather_f32(ctypes.byref(desc), _lib.ptr(store), _lib.ptr(pts), 0, 0.0, None, _lib.stream_ptr()) == 0
