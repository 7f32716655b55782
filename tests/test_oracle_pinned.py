"""CPU: pin the oracle (oracle/pvamd_oracle.c) against everything available without the reference's dependencies:
golden vectors produced by running the liftable parts of the reference (tests/golden/make_golden.py), closed-form
SDFs, and the property assertions of the reference's own tests.  Third-party arithmetic stays UNPINNED."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from pytorch_volumetric_amd import mesh_io
from tests import helpers as H

G = np.load(os.path.join(H.GOLDEN, "reference_lifted.npz"))


def golden_grid(tag, oob_mode=1):
    rng = G[f"cached/{tag}/range_snapped"]
    dt = np.float64 if tag == "f64" else np.float32
    return oracle.Grid(G[f"cached/{tag}/val_grid"], G[f"cached/{tag}/grad_grid"], rng[:, 0].astype(dt),
                       rng[:, 1].astype(dt), G[f"cached/{tag}/bb"], oob_mode=oob_mode)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_index_arithmetic_matches_reference_glue(tag):
    """Keys / validity as computed by the reference's CachedSDF code over the (shimmed) value-range view."""
    key, flat, valid = oracle.voxel_index(golden_grid(tag), G[f"cached/{tag}/points"])
    assert np.array_equal(key, G[f"cached/{tag}/keys"])
    assert np.array_equal(valid, G[f"cached/{tag}/valid"])
    assert 0.2 < valid.mean() < 0.8


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_cached_query_matches_reference_glue(tag):
    """sdf.py:535-571 executed verbatim (make_golden.py group B) vs the oracle: in-range values are gathers and must be
    bit-identical; the BOUNDING_BOX branch differs only by torch's norm rounding (<= 1 ulp)."""
    val, grad, oob = oracle.cached_query(golden_grid(tag), G[f"cached/{tag}/points"])
    rv, rg = G[f"cached/{tag}/val"], G[f"cached/{tag}/grad"]
    inb = ~oob
    assert np.array_equal(val[inb], rv[inb]) and np.array_equal(grad[inb], rg[inb])
    finite = oob & np.isfinite(rg).all(axis=1)
    assert np.abs(val[oob] - rv[oob]).max() <= 1.2e-7
    assert np.abs(grad[finite] - rg[finite]).max() <= 2.4e-7
    assert np.array_equal(np.isnan(grad), np.isnan(rg))  # 0/0 inside the box, like the reference
    out = oracle.cached_outside(golden_grid(tag), G[f"cached/{tag}/points"], 0.02)
    assert np.array_equal(out, G[f"cached/{tag}/outside"])


def test_oracle_composed_query_matches_reference_glue():
    """sdf.py:392-433 executed verbatim.  The reference transforms with a torch matmul and rotates normals with a
    general matrix inverse, so values agree to fp32 round-off except for the few points that a 1-ulp coordinate
    difference moves across a voxel boundary."""
    g = golden_grid("f64")
    pts = G["composed/points"]
    for name, A in (("single", 1), ("batched", 4)):
        tf = G[f"composed/{name}/tf"]
        val, grad, leaf = oracle.composed_query([g, g, g], tf, A, pts)
        rv, rg = G[f"composed/{name}/val"], G[f"composed/{name}/grad"]
        if A == 1:
            assert rv.shape == (1500,) and rg.shape == (1500, 3)  # the reference returns FLAT results here
            rv, rg = rv[None], rg[None]
        else:
            assert rv.shape == (4, 3, 500)
            rv, rg = rv.reshape(A, -1), rg.reshape(A, -1, 3)
        close = np.isclose(val, rv, rtol=0, atol=1e-6)
        assert close.mean() > 0.995, f"{(~close).sum()} of {close.size} values differ"
        gclose = np.isclose(grad[close], rg[close], rtol=0, atol=2e-6) | np.isnan(rg[close])
        assert gclose.mean() > 0.999


def test_cube_closed_form():
    m = mesh_io.load_mesh(H.mesh_path("box_template.obj"))  # the reference's own cube [-1,1]^3
    om = oracle.Mesh(m.triangle_soup(), m.triangle_normals(), m.aabb()[1] + 1.0)
    pts = np.random.default_rng(0).uniform(-2.5, 2.5, (20000, 3)).astype(np.float32)
    c, d, g, f, n = oracle.mesh_query(om, pts, seed=3)
    q = np.abs(pts.astype(np.float64)) - 1
    ref = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
    assert np.abs(d - ref).max() < 1e-6
    assert np.abs(np.linalg.norm(c - pts, axis=1) - np.abs(ref)).max() < 1e-6
    # well inside one face's prism: next to an edge, the face and its neighbour across the edge tie in fp32 distance and
    # the lowest face id wins (tie-breaking is one of the unpinned third-party behaviours)
    face_region = ((q > 0).sum(axis=1) == 1) & (np.abs(ref) > 1e-2) & (np.sort(q, axis=1)[:, 1] < -1e-2)
    expect = np.sign(pts) * (q > 0)
    # the gradient is (closest - p)/d: a closest-point round-off of ~2e-7 shows up as ~2e-7/d
    assert (np.abs(g[face_region] - expect[face_region]).max(axis=1) < 3e-6 / np.abs(ref[face_region])).all()
    assert np.abs(n[face_region] - expect[face_region]).max() < 1e-6  # the closest face's normal


def test_sphere_mesh_close_to_analytic_and_sign_by_parity():
    r = 0.1
    m = mesh_io.uv_sphere_mesh(r, 64, 32)
    om = oracle.Mesh(m.triangle_soup(), m.triangle_normals(), m.aabb()[1] + 1.0)
    pts = np.random.default_rng(1).uniform(-0.2, 0.2, (3000, 3)).astype(np.float32)
    _, d, g, _, _ = oracle.mesh_query(om, pts, seed=0)
    ref = np.linalg.norm(pts, axis=1) - r
    assert np.abs(d - ref).max() < r * (1 - np.cos(np.pi / 32)) + 1e-6
    clear = np.abs(ref) > 2e-3
    assert ((d < 0) == (ref < 0))[clear].all()
    radial = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    assert (np.sum(g * radial, axis=1)[clear] > 0.99).all()  # gradient points towards increasing SDF


@pytest.mark.parametrize("mesh", ["probe.obj", "offset_wrench_nogrip.obj"])
def test_surface_samples_have_zero_distance(mesh):
    """tests/test_sdf.py:18-23 of the reference: |sdf| < 1e-4 on 1000 surface samples; gradient = face normal."""
    m = mesh_io.load_mesh(H.mesh_path(mesh))
    om = oracle.Mesh(m.triangle_soup(), m.triangle_normals(), m.aabb()[1] + 1.0)
    rng = np.random.default_rng(0)
    areas = m.triangle_areas()
    fid = rng.choice(len(areas), 1000, p=areas / areas.sum())
    r1, r2 = np.sqrt(rng.random(1000)), rng.random(1000)
    t = m.triangle_soup()[fid]
    pts = ((1 - r1)[:, None] * t[:, 0] + (r1 * (1 - r2))[:, None] * t[:, 1] + (r1 * r2)[:, None] * t[:, 2])
    _, d, g, f, _ = oracle.mesh_query(om, pts.astype(np.float32), seed=0)
    assert np.abs(d).max() < 1e-4
    assert (np.sum(g * m.triangle_normals()[fid], axis=1) > 0.999).mean() > 0.9


def test_jitter_is_counter_based_and_bounded():
    m = mesh_io.box_mesh()
    om = oracle.Mesh(m.triangle_soup(), m.triangle_normals(), [2.0, 2.0, 2.0])
    d0, d1 = oracle.jitter_dir(om, 5, 17), oracle.jitter_dir(om, 5, 17)
    assert np.array_equal(d0, d1)
    dirs = np.stack([oracle.jitter_dir(om, 5, i) for i in range(4000)])
    noise = (dirs.astype(np.float64) - 2.0) / 1e-4
    assert abs(noise.mean()) < 0.05 and 0.9 < noise.std() < 1.1 and np.abs(noise).max() <= 6.01
    assert not np.array_equal(oracle.jitter_dir(om, 6, 17), d0)


def test_chamfer_exact_pose_is_zero_and_translation_known_answer():
    """tests/test_chamfer.py:36-38: identity pose -> 0; a pure translation of a face-sampled cube cloud along the
    face normal by t gives mean (scale*t)^2 on that face."""
    m = mesh_io.box_mesh()
    om = oracle.Mesh(m.triangle_soup(), m.triangle_normals(), [2.0, 2.0, 2.0])
    rng = np.random.default_rng(2)
    pts = np.concatenate([rng.uniform(-0.8, 0.8, (500, 2)), np.ones((500, 1))], axis=1).astype(np.float32)  # top face
    W = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    W[1, 2, 3] = 0.05
    out = oracle.chamfer_mesh(om, W, pts, scale=1000.0) / len(pts)
    assert out[0] < 1e-4
    assert abs(out[1] - 2500.0) < 1e-2


def test_transform_stack_matches_float64_algebra():
    rng = np.random.default_rng(3)
    S, A = 3, 5
    off = H.random_rigid(S, seed=1).numpy()
    world = H.random_rigid(S * A, seed=2, trans=1.0).numpy()
    out = oracle.transform_stack(np.linalg.inv(off.astype(np.float64)).astype(np.float32), world, S, A)
    ref = np.einsum("sij,sajk->saik", np.linalg.inv(off.astype(np.float64)),
                    np.linalg.inv(world.astype(np.float64)).reshape(S, A, 4, 4)).reshape(-1, 4, 4)
    assert np.abs(out - ref).max() < 2e-6


def test_torch_opforop_baseline_agrees_with_the_c_oracle():
    """The op-for-op torch restatement timed by bench.py (B1) and the C oracle (B2) compute the same thing."""
    from oracle.torch_opforop import CachedOpForOp
    for tag in ("f64", "f32"):
        rng = G[f"cached/{tag}/range_snapped"]
        dt = torch.float64 if tag == "f64" else torch.float32
        ref = CachedOpForOp(torch.from_numpy(G[f"cached/{tag}/val_grid"]), torch.from_numpy(G[f"cached/{tag}/grad_grid"]),
                            torch.tensor(rng[:, 0], dtype=dt), torch.tensor(rng[:, 1], dtype=dt),
                            torch.from_numpy(G[f"cached/{tag}/bb"]))
        pts = torch.from_numpy(G[f"cached/{tag}/points"])
        v, g = ref(pts)
        # identical to what the reference's own code produced (make_golden.py group B) ...
        assert np.array_equal(v.numpy(), G[f"cached/{tag}/val"])
        assert np.array_equal(g.numpy(), G[f"cached/{tag}/grad"], equal_nan=True)
        # ... and to the C oracle up to torch's norm rounding in the out-of-range branch
        ov, og, oob = oracle.cached_query(golden_grid(tag), pts.numpy())
        assert np.array_equal(v.numpy()[~oob], ov[~oob]) and np.abs(v.numpy() - ov).max() <= 1.2e-7


def pin_status():
    """Which third-party arithmetic the committed vectors were generated with (tests/golden/make_golden.py writes the
    booleans): the REAL multidim_indexing view / pytorch_kinematics Transform3d / open3d scene, or the shims."""
    return {k: bool(G[f"pinned/{k}"]) for k in ("view", "transform", "embree")}


def test_pin_status_is_recorded_reported_and_consistent():
    """Re-pinning is a re-run of make_golden.py where the packages exist; this test says what the committed vectors pin and
    keeps the UNPINNED banner in place until all three are real."""
    from pytorch_volumetric_amd import voxel
    status = pin_status()
    print("third-party arithmetic pinned by the real packages:", status)
    if status["view"]:
        assert int(G["pinned/index_rule"]) == voxel.INDEX_RULE  # the detected rule of the real view is the one in force
    root = os.path.dirname(H.GOLDEN)
    header = open(os.path.join(os.path.dirname(root), "oracle", "pvamd_oracle.c")).read()[:4000]
    if not all(status.values()):
        assert "UNPINNED" in header.upper()
    if status["embree"]:
        assert "mesh/points" in G.files


@pytest.mark.skipif("mesh/points" not in G.files, reason="no open3d where the vectors were generated: Embree arithmetic unpinned")
def test_oracle_mesh_query_matches_the_reference_over_open3d():
    """Only once make_golden.py ran with open3d: the oracle's closest point / distance / gradient / normal against the
    reference's own _do_object_frame_closest_point (sdf.py:122-172) on the drill; points whose sign depends on the jitter
    seed (recorded by the generator) are compared by magnitude only."""
    obj = H.oracle_mesh_from_factory(__import__("workloads").build_drill())
    closest, dist, grad, _, _ = oracle.mesh_query(obj, G["mesh/points"], seed=0)
    stable = ~G["mesh/unstable"]
    assert np.allclose(closest, G["mesh/closest"], atol=1e-6)
    assert np.allclose(np.abs(dist), np.abs(G["mesh/distance"]), atol=1e-6)
    assert np.array_equal(np.sign(dist[stable]), np.sign(G["mesh/distance"][stable]))
    far = np.abs(G["mesh/distance"]) > 2e-3
    assert np.allclose(grad[stable & far], G["mesh/gradient"][stable & far], atol=1e-4)
