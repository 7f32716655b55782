"""-m gpu: the reference's largest test configuration at its own size -- tests/test_model_to_sdf.py:263-326
(`test_single_link_robot`): offset_wrench.urdf, cache_link_sdf_factory(resolution=0.001, padding=0.05) = a 218 x 126 x 111
= 3,048,948-voxel cache of the NON-watertight offset_wrench_nogrip.obj, every grid point of the padded surface box queried,
the near-surface selection re-queried flat and batched under a configuration batch.

Statement for statement the reference's test (headless: no open3d point cloud), plus what it cannot check itself: the cache
the mesh kernel built against the CPU oracle, bit for bit, on a 24,000-voxel slice, under two jitter seeds -- on an open mesh
the ray-parity sign of a voxel can depend on the jitter, so the centres whose sign changes between the seeds
are counted and reported (the GPU must still equal the oracle under each seed)."""
import os
import time

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

RES, PAD = 0.001, 0.05
GRID = (218, 126, 111)  # computed from the reference's own grid functions


@pytest.fixture(scope="module")
def wrench(tmp_path_factory):
    urdf = open(H.mesh_path("offset_wrench.urdf")).read()
    chain = pv.build_serial_chain_from_urdf(urdf, "offset_wrench")
    cache = str(tmp_path_factory.mktemp("wrench") / "sdf_cache.pkl")  # the reference's default is a pickle in the cwd
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sdf = pv.RobotSDF(chain, path_prefix=H.MESHES,
                      link_sdf_cls=pv.cache_link_sdf_factory(resolution=RES, padding=PAD, device="cuda", cache_path=cache))
    torch.cuda.synchronize()
    return sdf, cache, time.perf_counter() - t0


def test_single_link_robot_at_the_reference_size(wrench):
    sdf, cache, build_s = wrench
    leaf = sdf.sdf.sdfs[0]
    assert tuple(leaf._view.shape) == GRID and leaf._packed.shape == (3_048_948, 4)
    d = "cuda"
    th = torch.zeros(6, device=d)  # trans (0, 0, 0), rot (0, 0, 0): test_model_to_sdf.py:273-282
    sdf.set_joint_configuration(th.view(1, -1))
    query_range = sdf.surface_bounding_box(padding=0.05)[0]
    coords, pts = pv.get_coordinates_and_points_in_grid(0.001, query_range, device=d)
    assert pts.shape[0] > 2_900_000
    sdf_val, sdf_grad = sdf(pts)
    assert sdf_val.shape == (1, pts.shape[0]) and sdf_grad.shape == (1, pts.shape[0], 3)  # th was (1, 6): a batch of 1
    sdf_val, sdf_grad = sdf_val[0], sdf_grad[0]
    near_surface = sdf_val.abs() < 0.001
    surf_pts = pts[near_surface]
    surf_norms = sdf_grad[near_surface]
    assert surf_pts.shape[0] > 10_000
    assert torch.allclose(surf_norms.norm(dim=-1), torch.ones_like(surf_norms[:, 0]), atol=1e-4)

    # multiple joint configurations (test_model_to_sdf.py:301-308)
    B = 5
    th = th.view(1, -1).repeat(B, 1)
    sdf.set_joint_configuration(th)
    query_range = sdf.surface_bounding_box(padding=0.05)
    assert query_range.shape == (B, 3, 2)
    for i in range(1, B):
        assert torch.equal(query_range[0], query_range[i])

    # non-batch query under a batch of configurations (:310-319)
    BB, N = 10, 100
    assert surf_pts.shape[0] > BB * N
    test_pts = surf_pts[:BB * N]
    sdf_vals, sdf_grads = sdf(test_pts)
    assert sdf_vals.shape == (B, BB * N) and sdf_grads.shape == (B, BB * N, 3)
    assert torch.allclose(sdf_vals.abs(), torch.zeros_like(sdf_vals), atol=1e-3)

    # batch query under a batch of configurations (:321-326)
    batch_pts = test_pts.view(BB, N, 3)
    batch_sdf_vals, batch_sdf_grads = sdf(batch_pts)
    assert batch_sdf_vals.shape == (B, BB, N) and batch_sdf_grads.shape == (B, BB, N, 3)
    assert torch.equal(batch_sdf_vals, sdf_vals.view(B, BB, N))
    assert torch.equal(batch_sdf_grads.nan_to_num(9.0), sdf_grads.view(B, BB, N, 3).nan_to_num(9.0))
    print(f"\nwrench res 0.001 pad 0.05: RobotSDF built (mesh load + {np.prod(GRID)} voxel cache + 49 MB pickle) in {build_s:.2f} s; "
          f"{int(near_surface.sum())} of {pts.shape[0]} grid points within 1 mm of the surface")


def test_the_pickled_cache_is_the_reference_layout_and_reloads_bit_exact(wrench):
    sdf, cache, _ = wrench
    leaf = sdf.sdf.sdfs[0]
    data = torch.load(cache, weights_only=False)
    (name, (val, grad)), = data.items()
    assert name == leaf.name and tuple(val.shape) == GRID and tuple(grad.shape) == (3_048_948, 3)  # sdf.py:504-505,515
    obj = leaf.gt_sdf.obj_factory

    class NeverQueried(pv.MeshSDF):  # the reference needs gt_sdf for the surface box (sdf.py:525) even when the cache loads
        def __call__(self, pts):
            raise AssertionError("the cache entry was not used")

    again = pv.CachedSDF(obj.name, RES, obj.bounding_box(padding=PAD), NeverQueried(obj), device="cuda", cache_path=cache)
    assert again.name == name and torch.equal(again._packed, leaf._packed)


@pytest.mark.parametrize("seed", [0, 1])
def test_cache_slice_matches_the_oracle_bitwise_under_two_jitter_seeds(wrench, seed, record_property):
    """24,000 voxel centres (three runs of 8,000 consecutive ones: first x-slab, the slab through the handle, the last) of the
    cache the mesh kernel fills, against the oracle's double loop: value, gradient, bit for bit."""
    sdf, _, _ = wrench
    leaf = sdf.sdf.sdfs[0]
    obj = leaf.gt_sdf.obj_factory
    coords, _ = pv.get_coordinates_and_points_in_grid(RES, leaf.ranges, get_points=False)
    n_all = int(np.prod(GRID))
    omesh = H.oracle_mesh_from_factory(obj)
    old = obj.jitter_seed
    try:
        obj.jitter_seed = seed
        flips = total = 0
        for start in (0, (n_all // 2 // 8000) * 8000, n_all - 8000):
            idx = torch.arange(start, start + 8000)
            ix, iy, iz = idx // (GRID[1] * GRID[2]), (idx // GRID[2]) % GRID[1], idx % GRID[2]
            centres = torch.stack((coords[0][ix], coords[1][iy], coords[2][iz]), dim=1)
            # the build queries ALL centres in one call: a centre's jitter is a function of its global index
            res = obj.object_frame_closest_point(centres.cuda(), index_base=start)
            _, od, og, _, _ = oracle.mesh_query(omesh, centres.numpy(), seed=seed, index_base=start)
            assert np.array_equal(res.distance.cpu().numpy(), od), "signed distance differs from the oracle"
            assert np.array_equal(res.gradient.cpu().numpy(), og, equal_nan=True)
            if seed == 0:  # the cache itself was built under seed 0
                assert np.array_equal(leaf._packed[start:start + 8000, 0].cpu().numpy(), od)
                assert np.array_equal(leaf._packed[start:start + 8000, 1:].cpu().numpy(), og, equal_nan=True)
            else:
                flips += int((np.sign(leaf._packed[start:start + 8000, 0].cpu().numpy()) != np.sign(od)).sum())
                total += 8000
        if seed != 0:
            record_property("sign_unstable_voxel_centres", f"{flips} of {total}")
            print(f"\nopen mesh: {flips} of {total} voxel centres change sign between jitter seeds 0 and {seed}")
            assert flips < 0.02 * total  # an open mesh may flip some; a kernel bug flips half
    finally:
        obj.jitter_seed = old
