"""-m gpu: fused ComposedSDF / RobotSDF kernel vs the CPU oracle (BASELINE configs C3, C4), bit-exact."""
import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def make_leaf(f64=True, res=0.01, padding=0.1):
    gt = H.drill_like_gt()
    return pv.CachedSDF("drill_like", res, H.padded_range(H.DRILL_BB, padding, as_numpy=f64), gt, device="cuda",
                        cache_path=None)


def scene_points(n, seed, extent=0.5):
    return H.uniform_points(n, [-extent] * 3, [extent] * 3, seed)


@pytest.mark.parametrize("S,A,P", [(1, 1, 17), (2, 1, 1000), (8, 1, 40_000), (8, 5, 4096), (3, 7, 1001)])
def test_composed_matches_oracle_bitwise(S, A, P):
    leaves = [make_leaf(f64=(s % 2 == 0)) for s in range(S)]  # mixed float64 / float32 index arithmetic
    tfm = H.random_rigid(S * A, seed=S * 100 + A)
    comp = pv.ComposedSDF(leaves, pv.Transform3d(matrix=tfm))
    if A > 1:
        comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    pts = scene_points(P, seed=P)
    val, grad = comp(pts.cuda())
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    oval, ograd, oleaf = oracle.composed_query(ogrids, tfm.numpy(), A, pts.numpy())
    if A == 1:
        assert val.shape == (P,) and grad.shape == (P, 3)  # flat without a transform batch (sdf.py:418-433)
        oval, ograd = oval[0], ograd[0]
    else:
        assert val.shape == (A, P) and grad.shape == (A, P, 3)
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    assert len(np.unique(oleaf)) == min(S, len(np.unique(oleaf)))  # sanity: argmin ran


def test_translation_only_compose_is_min_of_shifted_leaves():
    """Known answer without any oracle (pattern of the reference's tests/test_sdf.py:61-80):
    sdf(p) = min(sdf1(p + t1), sdf2(p + t2)) for pure translations."""
    leaf = make_leaf()
    t1, t2 = torch.tensor([0.1, 0.0, 0.0]), torch.tensor([-0.2, 0.0, 0.2])
    tsf = pv.Translate(*t1.tolist()).stack(pv.Translate(*t2.tolist()))
    comp = pv.ComposedSDF([leaf, leaf], tsf)
    pts = scene_points(20_000, seed=3, extent=0.35).cuda()
    val, grad = comp(pts)
    v1, g1 = leaf(pts + t1.cuda())
    v2, g2 = leaf(pts + t2.cuda())
    take2 = v2 < v1
    assert torch.equal(val, torch.where(take2, v2, v1))
    assert torch.equal(grad.nan_to_num(9.0), torch.where(take2.unsqueeze(-1), g2, g1).nan_to_num(9.0))


def test_config_batch_equals_per_config_loop_and_shapes():
    """tests/test_model_to_sdf.py:173-212 and :301-326 of the reference, headless."""
    S, A = 4, 6
    leaves = [make_leaf() for _ in range(S)]
    tfm = H.random_rigid(S * A, seed=42).reshape(S, A, 4, 4)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm.reshape(-1, 4, 4)), batch_dim=(2, 3))
    pts = scene_points(5 * 64, seed=8).reshape(5, 64, 3).cuda()
    val, grad = comp(pts)
    assert val.shape == (2, 3, 5, 64) and grad.shape == (2, 3, 5, 64, 3)
    flat_val, flat_grad = comp(pts.reshape(-1, 3))
    assert torch.equal(flat_val.reshape(val.shape), val)
    for a in range(A):
        single = pv.ComposedSDF(leaves, pv.Transform3d(matrix=tfm[:, a]))
        v, g = single(pts.reshape(-1, 3))
        assert torch.equal(v, val.reshape(A, -1)[a])
        assert torch.equal(g.nan_to_num(9.0), grad.reshape(A, -1, 3)[a].nan_to_num(9.0))


def test_generic_path_with_uncached_leaves_agrees_with_fused_path():
    leaves = [make_leaf() for _ in range(3)]
    tfm = H.random_rigid(3, seed=5)

    class Opaque(pv.ObjectFrameSDF):  # hides the CachedSDF type -> forces the per-leaf path
        def __init__(self, inner):
            self.inner = inner

        def __call__(self, p):
            return self.inner(p)

        def surface_bounding_box(self, **kw):
            return self.inner.surface_bounding_box(**kw)

    fused = pv.ComposedSDF(leaves, pv.Transform3d(matrix=tfm))
    generic = pv.ComposedSDF([Opaque(l) for l in leaves], pv.Transform3d(matrix=tfm))
    pts = scene_points(10_000, seed=2).cuda()
    v1, g1 = fused(pts)
    v2, g2 = generic(pts)
    # torch's matmul rounds differently from the kernel's fma chain: 1e-5 everywhere except where that last-place
    # difference crosses a half-voxel plane or a range edge in some leaf frame -- every such point is accounted for
    rep = {}
    n_bad, n_unexplained = H.composed_disagreements_explained(leaves, tfm.numpy(), 1, pts.cpu().numpy(),
                                                              v1.cpu().numpy(), v2.cpu().numpy(), atol=1e-5, report=rep)
    print(f"generic vs fused: {n_bad} of {v1.numel()} values differ by more than 1e-5, all within {rep['max_units_needed']:.2f} "
          "rounding units (bound: 8) of a voxel / range boundary")
    assert n_unexplained == 0 and n_bad < 0.01 * v1.numel()
    close = torch.isclose(v1, v2, atol=1e-5)
    assert torch.allclose(g1[close].nan_to_num(0.), g2[close].nan_to_num(0.), atol=1e-4)


def test_full_size_c3_properties():
    """BASELINE C3 at full size (8 leaves, 4M points): determinism + slice vs oracle + leaf-permutation invariance
    of the minimum value."""
    S = 8
    leaf = make_leaf()
    tfm = H.random_rigid(S, seed=0)
    comp = pv.ComposedSDF([leaf] * S, pv.Transform3d(matrix=tfm))
    pts = scene_points(4_000_000, seed=0).cuda()
    v1, g1 = comp(pts)
    v2, _ = comp(pts)
    assert torch.equal(v1, v2)
    rev = pv.ComposedSDF([leaf] * S, pv.Transform3d(matrix=tfm.flip(0)))
    v3, _ = rev(pts)
    assert torch.equal(v1, v3)  # min over leaves does not depend on their order
    og = H.oracle_grid_from_cached(leaf)
    oval, ograd, _ = oracle.composed_query([og] * S, tfm.numpy(), 1, pts[:50_000].cpu().numpy())
    assert np.array_equal(v1[:50_000].cpu().numpy(), oval[0], equal_nan=True)
    assert np.array_equal(g1[:50_000].cpu().numpy(), ograd[0], equal_nan=True)


def test_leaf_culling_is_exact_on_coherent_and_far_queries():
    """Grid-ordered points (whole waves far from most leaves -> leaves skipped) and points far outside every leaf must
    still match the oracle bit for bit; so must a scene where a leaf's grid range is SMALLER than its bounding box."""
    S, A = 8, 10
    leaves = [make_leaf(f64=(s % 2 == 0), padding=0.05) for s in range(S)]
    tfm = H.random_rigid(S * A, seed=77, trans=1.5)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    _force_flags(comp, 4)  # the wave-tile kernel (the one that culls), whatever the entry point would pick at this size
    ax = torch.linspace(-2.0, 2.0, 48)
    pts = torch.cartesian_prod(ax, ax, ax)[: 48 * 48 * 48 // 256 * 256]
    far = H.uniform_points(2048, [30.0] * 3, [40.0] * 3, seed=1)
    pts = torch.cat((pts, far))
    val, grad = comp(pts.cuda())
    oval, ograd, oleaf = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    # range box inside the bounding box (negative padding): the out-of-range value can be 0 with a NaN gradient
    gt = H.drill_like_gt()
    tight = pv.CachedSDF("tight", 0.01, H.padded_range(H.DRILL_BB, -0.02), gt, device="cuda", cache_path=None)
    comp2 = pv.ComposedSDF([tight, leaves[0]], pv.Transform3d(matrix=tfm[:2]))
    q = H.uniform_points(1 << 20, [-2.0] * 3, [2.0] * 3, seed=5)
    v2, g2 = comp2(q.cuda())
    ov, og, _ = oracle.composed_query([H.oracle_grid_from_cached(tight), H.oracle_grid_from_cached(leaves[0])],
                                      tfm[:2].numpy(), 1, q.numpy())
    assert np.array_equal(v2.cpu().numpy(), ov[0], equal_nan=True)
    assert np.array_equal(g2.cpu().numpy(), og[0], equal_nan=True)


def test_more_leaves_than_the_culling_table_holds():
    """S = 70 > 64: leaves beyond the LDS culling table take the un-culled branch of the same loop (wave-tile kernel:
    257 tiles x 16 configurations, the last tile with 4 points)."""
    S, A = 70, 16
    leaf = make_leaf(res=0.02)
    tfm = H.random_rigid(S * A, seed=9, trans=0.8)
    comp = pv.ComposedSDF([leaf] * S, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    _force_flags(comp, 4)
    pts = scene_points(65_536 + 4, seed=4, extent=1.0)
    val, grad = comp(pts.cuda())
    og = H.oracle_grid_from_cached(leaf)
    oval, ograd, oleaf = oracle.composed_query([og] * S, tfm.numpy(), A, pts.numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    assert oleaf.max() >= 64  # the late leaves do win somewhere


def test_compose_of_mesh_sdfs_like_the_reference_test():
    """tests/test_sdf.py:61-80 of the reference: two drills (MeshSDF leaves) placed with pure translations."""
    obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
    sdf1, sdf2 = pv.MeshSDF(obj), pv.MeshSDF(obj)
    tsf = pv.Translate(0.1, 0, 0).stack(pv.Translate(-0.2, 0, 0.2))
    comp = pv.ComposedSDF([sdf1, sdf2], tsf)
    _, pts = pv.get_coordinates_and_points_in_grid(0.002, obj.bounding_box(0.01))
    pts = pts[torch.randperm(len(pts), generator=torch.Generator().manual_seed(0))[:1000]].cuda()
    vals, grads = comp(pts)
    assert vals.shape == (1000,) and grads.shape == (1000, 3)
    v1, g1 = sdf1(pts + torch.tensor([0.1, 0.0, 0.0], device="cuda"))
    v2, g2 = sdf2(pts + torch.tensor([-0.2, 0.0, 0.2], device="cuda"))
    # the jitter is indexed by point number, so the per-leaf queries above see the same rays as inside the composition
    expect = torch.minimum(v1, v2)
    assert torch.allclose(vals, expect, atol=1e-6)
    assert torch.allclose(grads, torch.where((v2 < v1).unsqueeze(-1), g2, g1), atol=1e-5)


@pytest.mark.parametrize("flags", [4, 4 | 16, 4 | 1, 2])
def test_both_index_modes_give_the_reference_index_where_the_estimate_is_shaky(flags):
    """pvamd_composed_query's tuning hints must never change a result: wave-tile kernel with flagged points redone after the
    loop (4), with the exact statements inline (4 | 1), and the per-lane kernel (2).  A leaf far from its own origin (coordinates ~200 x the resolution -> a wide error bound on the
    fp32 index estimate) and query points sprayed on its half-voxel planes make flagged visits the rule."""
    gt = H.AnalyticEllipsoidSDF([7.0, -5.0, 3.0], [0.3, 0.2, 0.25], [[6.7, 7.3], [-5.2, -4.8], [2.75, 3.25]])
    rng = [(6.5, 7.5), (-5.5, -4.5), (2.5, 3.5)]
    leaves = [pv.CachedSDF(f"far{i}", 0.02, np.array(rng) if i % 2 == 0 else rng, gt, device="cuda", cache_path=None)
              for i in range(4)]
    S, A = 4, 24
    tfm = H.random_rigid(S * A, seed=11, trans=0.2)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    v = leaves[0]._view
    g = np.random.default_rng(5)
    n = 1 << 15
    k = g.integers(0, np.array(v.shape) - 1, size=(n, 3))
    on_plane = v.dmin.numpy() + (k + 0.5) * v.dres.numpy() + g.normal(scale=2e-6, size=(n, 3))
    # bring the leaf-frame targets back to the object frame of leaf (s = i mod S, a = 0) so that they land on the planes
    inv = tf_inv = torch.linalg.inv(tfm.reshape(S, A, 4, 4)[:, 0].double()).numpy()
    pts = np.stack([inv[i % S, :3, :3] @ on_plane[i] + inv[i % S, :3, 3] for i in range(n)]).astype(np.float32)
    pts = torch.from_numpy(pts)
    dev = torch.device("cuda", torch.cuda.current_device())
    comp._leaf_grids(dev)
    comp._query_flags = flags
    val, grad = comp(pts.cuda())
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def test_mesh_leaves_with_rotations_and_a_batch_match_the_oracle_composition_bitwise():
    """ComposedSDF over MeshSDF leaves (the per-leaf path: pvamd_transform_points -> mesh kernel -> pvamd_compose_merge)
    against the same composition stated with the oracle's pieces: transform (fma chain), brute-force mesh query with the
    same jitter counter, merge.  Rigid transforms with rotations, a configuration batch of 3."""
    objs = [pv.MeshObjectFactory(H.mesh_path("probe.obj")), pv.MeshObjectFactory(H.mesh_path("box_template.obj"), scale=0.03)]
    S, A, P = 2, 3, 4000
    tfm = H.random_rigid(S * A, seed=21, trans=0.05)
    comp = pv.ComposedSDF([pv.MeshSDF(o) for o in objs], None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    pts = H.uniform_points(P, [-0.1] * 3, [0.1] * 3, seed=8)
    val, grad = comp(pts.cuda())
    assert val.shape == (A, P) and grad.shape == (A, P, 3)
    best_v, best_g = np.empty((A, P), np.float32), np.empty((A, P, 3), np.float32)
    best_leaf = np.empty((A, P), np.int32)
    for s, obj in enumerate(objs):
        tf_s = tfm.reshape(S, A, 4, 4)[s].numpy()
        x = oracle.transform_points(tf_s, pts.numpy())  # [A,P,3]
        _, d, g, _, _ = oracle.mesh_query(H.oracle_mesh_from_factory(obj), x.reshape(-1, 3), seed=obj.jitter_seed)
        oracle.compose_merge(tf_s, d.reshape(A, P), g.reshape(A, P, 3), s, best_v, best_g, best_leaf)
    assert np.array_equal(val.cpu().numpy(), best_v)
    assert np.array_equal(grad.cpu().numpy(), best_g, equal_nan=True)
    assert len(np.unique(best_leaf)) == 2


def test_non_rigid_transforms_take_the_general_path_and_stay_consistent():
    """A scaled (non-rigid) obj->leaf transform: the reference inverts it with a general matrix inverse (sdf.py:380) and
    its __call__ is valid for any affine map; here set_transforms detects it, leaves the fused kernel (whose culling and
    R^T need rigidity) and x = L p + t, g_obj = L^T g_leaf are applied per leaf."""
    leaf = make_leaf()
    m = H.random_rigid(2, seed=4, trans=0.1)
    m[1, :3, :3] *= 1.3  # uniform scale on the second leaf
    comp = pv.ComposedSDF([leaf, leaf], m)
    assert comp._rigid is False and not comp._fusable()
    inv = comp.link_frame_to_obj_frame[1].get_matrix()[0]
    assert torch.allclose(inv @ m[1], torch.eye(4), atol=1e-5)  # a true inverse, not R^T
    pts = scene_points(20_000, seed=12, extent=0.3).cuda()
    val, grad = comp(pts)
    v0, g0 = leaf(pts @ m[0, :3, :3].T.cuda() + m[0, :3, 3].cuda())
    v1, g1 = leaf(pts @ m[1, :3, :3].T.cuda() + m[1, :3, 3].cuda())
    expect = torch.minimum(v0, v1)
    near_tie = (v0 - v1).abs() < 1e-5
    assert torch.allclose(val[~near_tie], expect[~near_tie], atol=2e-5)  # torch matmul vs fma chain: last-place differences
    g_expect = torch.where((v1 < v0).unsqueeze(-1), g1 @ m[1, :3, :3].cuda(), g0 @ m[0, :3, :3].cuda())
    ok = torch.isclose(val, expect, atol=1e-6) & ~near_tie
    assert torch.allclose(grad[ok].nan_to_num(0.0), g_expect[ok].nan_to_num(0.0), atol=1e-4) and ok.float().mean() > 0.95
    rigid = pv.ComposedSDF([leaf, leaf], H.random_rigid(2, seed=4, trans=0.1))
    assert rigid._rigid is True and rigid._fusable()


def _force_flags(comp, flags):
    comp._leaf_grids(torch.device("cuda", torch.cuda.current_device()))
    comp._query_flags = flags


@pytest.mark.parametrize("A,P,flags", [(200, 15_251, 0), (200, 15_251, 4), (20, 15_251, 4), (20, 15_251, 0), (24, 20_481, 4 | 1), (5, 255, 4),
                                       (5, 257, 4), (3, 3, 4), (7, 513, 4), (2, 1, 4)])
def test_any_point_count_goes_through_the_wave_tile_kernel_bitwise(A, P, flags):
    """The reference README's own query has M = 15,251 points (README.md:177-200): (A, P) rows that start at any dword,
    and a last tile of 147 points.  The wave-tile kernel takes them itself (16-byte stores at 4-byte aligned addresses,
    partial last tile in the kernel); round 2 sent every P % 4 != 0 to the one-point-per-lane kernel.  flags 4 forces the
    wave-tile kernel for the small cases; (200, 15251) and (20, 15251) take whatever the entry point picks."""
    S = 8
    leaves = [make_leaf(f64=(s % 2 == 0)) for s in range(S)]
    tfm = H.random_rigid(S * A, seed=A + P, trans=0.3)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    if flags:
        _force_flags(comp, flags)
    pts = scene_points(P, seed=P, extent=0.6)
    val, grad = comp(pts.cuda())
    assert val.shape == (A, P) and grad.shape == (A, P, 3)
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


@pytest.mark.parametrize("flags", [4, 2, 2 | 8])
def test_buffers_at_any_dword_address(flags):
    """points / out_val / out_grad that are only 4-byte aligned (views into larger buffers), odd P, both kernels; the
    floats around the outputs must stay untouched."""
    S, A, P = 4, 6, 1000 + 3
    leaves = [make_leaf() for _ in range(S)]
    tfm = H.random_rigid(S * A, seed=3, trans=0.2)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    _force_flags(comp, flags)
    pts = scene_points(P, seed=1, extent=0.4)
    pbuf = torch.zeros(3 * P + 8, device="cuda")
    pbuf[1:1 + 3 * P] = pts.reshape(-1).cuda()
    pview = pbuf[1:1 + 3 * P].view(P, 3)
    vbuf = torch.full((A * P + 8,), -7.0, device="cuda")
    gbuf = torch.full((3 * A * P + 8,), -7.0, device="cuda")
    vview, gview = vbuf[3:3 + A * P].view(A, P), gbuf[1:1 + 3 * A * P].view(A, P, 3)
    assert pview.data_ptr() % 16 == 4 and vview.data_ptr() % 16 == 12 and gview.data_ptr() % 16 == 4
    comp.query_into(pview, vview, gview)
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.numpy())
    assert np.array_equal(vview.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(gview.cpu().numpy(), ograd, equal_nan=True)
    assert (vbuf[:3] == -7).all() and (vbuf[3 + A * P:] == -7).all() and (gbuf[:1] == -7).all() and (gbuf[1 + 3 * A * P:] == -7).all()


@pytest.mark.parametrize("flags", [0, 2])
def test_seventy_thousand_configurations_go_out_in_slabs(flags):
    """pvamd_composed_query carries the configuration in a grid dimension (<= 65535): A = 70,000 x P = 8 crosses the slab
    border in both kernels (the wave-tile kernel with a last -- and only -- tile of 8 points); the reference takes any
    batch (sdf.py:370-383)."""
    S, A, P = 2, 70_000, 8
    leaves = [make_leaf(res=0.02) for _ in range(S)]
    tfm = H.random_rigid(S * A, seed=6, trans=0.3)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,))
    if flags:
        _force_flags(comp, flags)
    pts = scene_points(P, seed=2, extent=0.4)
    val, grad = comp(pts.cuda())
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves], tfm.numpy(), A, pts.numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)


def test_bucketed_and_packed_paths_take_more_than_65535_configurations():
    """The packed / bucketed entries carry the configuration in blockIdx.x, like the direct one: no 65,535 limit (round 2
    returned PVAMD_E_SHAPE and ShardedSDF fell back to its two-collective path)."""
    S, A, P = 2, 66_000, 300
    leaves = [make_leaf(res=0.02) for _ in range(S)]
    tfm = H.random_rigid(S * A, seed=16, trans=0.3)
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,), known_rigid=True)
    pts = scene_points(P, seed=12, extent=0.4).cuda()
    comp.bucket_points = True
    val, grad = comp(pts)
    comp.bucket_points = False
    v_direct, g_direct = comp(pts)
    assert torch.equal(val, v_direct) and torch.equal(grad.nan_to_num(3.0), g_direct.nan_to_num(3.0))
    pick = [0, 1, 65_534, 65_535, 65_536, A - 1]
    oval, ograd, _ = oracle.composed_query([H.oracle_grid_from_cached(l) for l in leaves],
                                           tfm.reshape(S, A, 4, 4)[:, pick].reshape(-1, 4, 4).numpy(), len(pick), pts.cpu().numpy())
    assert np.array_equal(val[pick].cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad[pick].cpu().numpy(), ograd, equal_nan=True)


def test_a_single_configuration_over_more_points_than_the_grid_has_lanes():
    """One configuration (the per-lane kernel) over 20 M points: its grid is capped at 65,535 workgroups of 256 lanes, so every
    lane takes a second round of the grid-stride loop -- the launch must equal the same points in chunks small enough for one
    round each, and the oracle on three runs."""
    import workloads
    leaves = [make_leaf(f64=(s % 2 == 0)) for s in range(3)]
    tfm = H.random_rigid(3, seed=77)
    comp = pv.ComposedSDF(leaves, pv.Transform3d(matrix=tfm))
    P = 20_000_003
    pts = workloads.uniform_points_device(P, [-0.5] * 3, [0.5] * 3, seed=5)
    val, grad = comp(pts)
    assert val.shape == (P,) and grad.shape == (P, 3)
    step = 4_000_000
    for a in range(0, P, step):
        v, g = comp(pts[a:a + step].contiguous())
        assert torch.equal(v, val[a:a + step]) and torch.equal(g.nan_to_num(7.0), grad[a:a + step].nan_to_num(7.0))
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    for a in (0, P // 2, P - 10_000):
        ov, og, _ = oracle.composed_query(ogrids, tfm.numpy(), 1, pts[a:a + 10_000].cpu().numpy())
        assert np.array_equal(val[a:a + 10_000].cpu().numpy(), ov[0], equal_nan=True)
        assert np.array_equal(grad[a:a + 10_000].cpu().numpy(), og[0], equal_nan=True)
