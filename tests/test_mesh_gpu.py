"""-m gpu: MeshSDF / object_frame_closest_point HIP kernel vs the CPU oracle and closed-form answers
(BASELINE config C1)."""
import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from pytorch_volumetric_amd import mesh_io
from tests import helpers as H

pytestmark = pytest.mark.gpu


def factory(name, **kw):
    return pv.MeshObjectFactory(H.mesh_path(name), **kw)


def grid_sample(obj, n, seed, res=0.002, pad=0.01):
    """The reference's own query pattern (tests/test_sdf.py:46-48): grid points in the padded box, random subset."""
    _, pts = pv.get_coordinates_and_points_in_grid(res, obj.bounding_box(pad))
    g = torch.Generator().manual_seed(seed)
    return pts[torch.randperm(len(pts), generator=g)[:n]]


def assert_query_matches(obj, pts, seed=0, oracle_points=None):
    """The GPU query over ALL of pts against the oracle -- over all of them too, or (oracle_points) over three runs of
    consecutive points at the start, in the middle and at the end (the oracle is a double loop; the jitter of a point is a
    function of its index, which the oracle is told)."""
    obj.jitter_seed = seed
    res = obj.object_frame_closest_point(pts.cuda(), compute_normal=True)
    face = obj._last_face_ids.cpu().numpy()
    got = (res.closest.cpu().numpy(), res.distance.cpu().numpy(), res.gradient.cpu().numpy(), face, res.normal.cpu().numpy())
    n = len(pts)
    runs = [(0, n)] if oracle_points is None or 3 * oracle_points >= n else \
        [(0, oracle_points), ((n - oracle_points) // 2, (n - oracle_points) // 2 + oracle_points), (n - oracle_points, n)]
    omesh = H.oracle_mesh_from_factory(obj)
    dist = []
    for a, b in runs:
        oc, od, og, of, on = oracle.mesh_query(omesh, pts[a:b].numpy(), seed=seed, index_base=a)
        assert np.array_equal(got[3][a:b], of), f"{(got[3][a:b] != of).sum()} face ids differ"
        assert np.array_equal(got[0][a:b], oc)
        assert np.array_equal(got[1][a:b], od), "signed distance (incl. ray-parity sign) differs"
        assert np.array_equal(got[2][a:b], og)
        assert np.array_equal(got[4][a:b], on)
        dist.append(od)
    return np.concatenate(dist)


@pytest.mark.parametrize("mesh,n", [("box_template.obj", 5000), ("probe.obj", 5000), ("offset_wrench_nogrip.obj", 3000)])
def test_mesh_query_matches_oracle_bitwise(mesh, n):
    obj = factory(mesh)
    bb = obj.bounding_box(padding_ratio=0.25)
    pts = H.uniform_points(n, bb[:, 0], bb[:, 1], seed=n)
    d = assert_query_matches(obj, pts, seed=7)
    assert (d < 0).any() and (d > 0).any()


@pytest.mark.parametrize("tile_split", [True, False])
def test_c1_drill_10k_grid_points_match_oracle(tile_split):
    """BASELINE config C1: MeshSDF on the YCB power drill (15,728 triangles), 10k grid query points.  With few points
    and many tiles the kernel spreads each point group's tiles over several workgroups (three launches meeting in a
    scratch buffer); with the scratch withheld it is the single-launch scan.  Same bits either way."""
    obj = factory("ycb_power_drill.npz")
    obj.tile_split = tile_split
    assert obj.num_faces == 15728
    pts = grid_sample(obj, 10_000, seed=0)
    d = assert_query_matches(obj, pts, seed=0)
    assert 0.05 < (d < 0).mean() < 0.9


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4100, 12_000, 16_500, 40_000, 131_072, 300_000, 524_288, 524_289])
def test_tile_split_path_for_every_group_count(n):
    """Partial last group, a single point, every tiles-per-block rule of the two-launch path (1 / 1.5 / 3 / groups / 512 tiles
    per block, four and two waves per block), counts on both sides of the 8192 groups where it hands over to the single launch,
    and every waves-per-group choice of the latter."""
    obj = factory("ycb_power_drill.npz")
    bb = obj.bounding_box(padding_ratio=0.3)
    pts = H.uniform_points(n, bb[:, 0], bb[:, 1], seed=100 + n)
    assert_query_matches(obj, pts, seed=n, oracle_points=40_000)


def test_cube_closed_form():
    """tests/pv_sdf_debug/box_template.obj is the cube [-1,1]^3: its SDF is closed-form."""
    obj = factory("box_template.obj")
    pts = H.uniform_points(20_000, [-2.5] * 3, [2.5] * 3, seed=1)
    val, grad = pv.MeshSDF(obj)(pts.cuda())
    q = pts.double().abs() - 1
    ref = torch.linalg.norm(q.clamp(min=0), dim=-1) + q.max(dim=-1).values.clamp(max=0)
    assert (val.cpu().double() - ref).abs().max() < 1e-6
    # gradient: unit length, equals the closed-form normal away from edges/medial axis
    far = (ref.abs() > 1e-2)
    assert torch.allclose(grad.cpu()[far].norm(dim=-1), torch.ones(int(far.sum())), atol=1e-5)
    face_region = ((q > 0).sum(dim=-1) == 1) & far & (q.sort(dim=-1).values[:, 1] < -1e-2)  # away from fp32 edge ties
    expect = torch.sign(pts) * (q > 0).float()
    assert ((grad.cpu()[face_region] - expect[face_region]).abs().amax(dim=-1).double() < 3e-6 / ref[face_region].abs()).all()


def _box_sdf(p, centre, half):
    q = (p - torch.tensor(centre, dtype=torch.float64)).abs() - torch.tensor(half, dtype=torch.float64)
    return torch.linalg.norm(q.clamp(min=0), dim=-1) + q.max(dim=-1).values.clamp(max=0)


@pytest.mark.parametrize("scene,z0", [("scene_mesh_separated.obj", 0.053449), ("scene_mesh_overlap.obj", -0.046147),
                                      ("scene_mesh_wrong.obj", 0.0), ("scene_mesh_gt.obj", 0.0)])
def test_reference_debug_scene_slices(scene, z0):
    """tests/pv_sdf_debug/test_export_composed_sdf.py of the reference, headless: the y = 0 slice (its query_range, through
    draw_sdf_slice) of MeshSDF over its four two-box scenes -- a slab B = [-0.3, 1.2] x [-0.4, 0.4] x [-0.2, 0] and a box
    A = [0.4, 1.2] x [-0.4, 0.4] x [z0, z0 + 0.2] above it: apart, overlapping, touching as two closed surfaces ("wrong": the
    shared face stays inside the solid) and touching as one proper union ("gt").  Bit for bit what the oracle's double loop gives,
    and the closed forms where they exist: outside both boxes every scene is min(box A, box B); the overlap of two closed
    surfaces reads as OUTSIDE (two ray hits: even parity, sdf.py:147-157); the doubled face of "wrong" pulls interior values
    to zero where "gt" keeps the true depth."""
    torch.manual_seed(0)  # draw_sdf_slice jitters the grid by 1e-6 from the global generator, as the reference does
    obj = factory(scene)
    sdf = pv.MeshSDF(obj)
    query_range = np.array([[-0.5, 2], [0, 0], [-0.2, 0.3]])
    val, grad, pts, ax, _, _, v = pv.draw_sdf_slice(sdf, query_range, device="cuda", do_plot=False)
    assert val.shape == (251 * 51,) and v.shape == (51, 251) and ax is None
    d = assert_query_matches(obj, pts.cpu(), seed=0)
    assert np.array_equal(d, val.cpu().numpy())
    p = pts.cpu().double()
    a = _box_sdf(p, (0.8, 0.0, z0 + 0.1), (0.4, 0.4, 0.1))
    b = _box_sdf(p, (0.45, 0.0, -0.1), (0.75, 0.4, 0.1))
    both = torch.minimum(a, b)
    got = val.cpu().double()
    clear = (a.abs() > 1e-4) & (b.abs() > 1e-4)  # a 1e-6 jitter decides the side of points on a face
    outside = (a > 0) & (b > 0) & clear
    assert outside.sum() > 5000 and (got[outside] - both[outside]).abs().max() < 1e-6
    in_a, in_b = (a < 0) & clear, (b < 0) & clear
    if scene == "scene_mesh_separated.obj":
        assert (got[clear] - both[clear]).abs().max() < 1e-6  # disjoint solids: the union's SDF everywhere
    elif scene == "scene_mesh_overlap.obj":
        inside_both = in_a & in_b
        assert inside_both.sum() > 100 and (got[inside_both] > 0).all()  # even parity: the reference's known artefact
        assert torch.allclose(got[inside_both], torch.minimum(a.abs(), b.abs())[inside_both], atol=1e-6)
        once = (in_a ^ in_b)
        assert (got[once] < 0).all() and torch.allclose(got[once].abs(), torch.minimum(a.abs(), b.abs())[once], atol=1e-6)
    else:
        inside = in_a | in_b
        assert (got[inside] < 0).all()
        if scene == "scene_mesh_wrong.obj":
            assert torch.allclose(got[inside], both[inside], atol=1e-6)  # every face counts, the buried one too
        else:
            assert (got[inside] <= both[inside] + 1e-6).all()
            near_seam = inside & (p[:, 0] > 0.6) & (p[:, 0] < 1.0) & (p[:, 2].abs() < 0.02)
            assert near_seam.sum() > 50 and (got[near_seam] < both[near_seam] - 0.05).all()  # true depth, no seam at z = 0


@pytest.mark.parametrize("mesh", ["probe.obj", "offset_wrench_nogrip.obj"])
def test_surface_points_have_zero_distance_and_batch_dims(mesh):
    """The reference's own assertions (tests/test_sdf.py:18-29)."""
    obj = factory(mesh)
    pts, normals, _ = pv.sample_mesh_points(obj, name=mesh, num_points=1000, dbpath=None)
    sdf = pv.MeshSDF(obj)
    vals, grads = sdf(pts)
    assert torch.allclose(vals.abs(), torch.zeros_like(vals), atol=1e-4)
    bvals, bgrads = sdf(pts.view(10, 100, -1))
    assert bvals.shape == (10, 100) and bgrads.shape == (10, 100, 3)
    assert torch.allclose(bvals.abs(), torch.zeros_like(bvals), atol=1e-4)
    # on the surface the gradient is the face normal (sdf.py:162-164)
    cos = (grads * normals).sum(-1)
    assert (cos > 0.999).float().mean() > 0.9


def test_scale_rotation_translation_of_the_mesh_frame():
    """sdf.py:104-113: scale, then rotate (xyzw quaternion), then translate by pos*scale."""
    s = 0.5
    quat_xyzw = (0.0, 0.0, np.sin(np.pi / 4), np.cos(np.pi / 4))  # 90 deg about z
    obj = factory("box_template.obj", scale=s, vis_frame_rot=quat_xyzw, vis_frame_pos=(1.0, 0.0, 0.0))
    bb = obj.bounding_box()
    assert np.allclose(bb, [[0.0, 1.0], [-0.5, 0.5], [-0.5, 0.5]], atol=1e-12)
    val, _ = pv.MeshSDF(obj)(torch.tensor([[0.5, 0.0, 0.0], [2.0, 0.0, 0.0]]))
    assert torch.allclose(val, torch.tensor([-0.5, 1.0]), atol=1e-6)


def test_a_collada_mesh_answers_like_the_same_triangles_from_an_obj(tmp_path):
    """.dae (the reference loads whatever open3d reads, sdf.py:104): the probe written as a COLLADA document whose node
    carries the identity gives the very bits of the .obj; with a node translation, the values of the shifted query."""
    src = mesh_io.load_mesh(H.mesh_path("probe.obj"))

    def write(path, translate):
        pos = " ".join(repr(float(x)) for x in src.vertices.reshape(-1))
        idx = " ".join(str(int(i)) for i in src.faces.reshape(-1))
        path.write_text(f"""<?xml version="1.0"?><COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
<asset><up_axis>Y_UP</up_axis></asset><library_geometries><geometry id="g"><mesh>
<source id="p"><float_array id="pa" count="{src.vertices.size}">{pos}</float_array>
<technique_common><accessor source="#pa" count="{len(src.vertices)}" stride="3"/></technique_common></source>
<vertices id="v"><input semantic="POSITION" source="#p"/></vertices>
<triangles count="{len(src.faces)}"><input semantic="VERTEX" source="#v" offset="0"/><p>{idx}</p></triangles>
</mesh></geometry></library_geometries><library_visual_scenes><visual_scene id="s"><node>
<translate>{translate[0]} {translate[1]} {translate[2]}</translate><instance_geometry url="#g"/></node></visual_scene>
</library_visual_scenes><scene><instance_visual_scene url="#s"/></scene></COLLADA>""")

    write(tmp_path / "probe.dae", (0, 0, 0))
    write(tmp_path / "probe_shifted.dae", (0.25, 0, -0.5))
    ref = factory("probe.obj")
    dae = pv.MeshObjectFactory("probe.dae", path_prefix=str(tmp_path))
    pts = H.uniform_points(4000, ref.bounding_box(0.02)[:, 0], ref.bounding_box(0.02)[:, 1], seed=8).cuda()
    a, b = ref.object_frame_closest_point(pts), dae.object_frame_closest_point(pts)
    assert torch.equal(a.distance, b.distance) and torch.equal(a.gradient, b.gradient) and torch.equal(a.closest, b.closest)
    shifted = pv.MeshObjectFactory("probe_shifted.dae", path_prefix=str(tmp_path))
    off = torch.tensor([0.25, 0.0, -0.5], device="cuda")
    c = shifted.object_frame_closest_point(pts + off)
    assert torch.allclose(c.distance, a.distance, atol=2e-6)


def test_voxel_view_and_filtered_points_of_a_mesh_and_of_its_cache():
    """ObjectFrameSDF.get_voxel_view / get_filtered_points (sdf.py:248-282) and CachedSDF.get_voxel_view (sdf.py:604-614) over the
    HIP path: the cube's interior voxel centres, a view addressed by coordinates that falls back to the SDF outside its grid."""
    obj = factory("box_template.obj", scale=0.1)  # the cube [-0.1, 0.1]^3
    sdf = pv.MeshSDF(obj)
    grid = pv.VoxelGrid(0.02, [(-0.2, 0.2)] * 3, device="cuda")
    inside = sdf.get_filtered_points(lambda v: v < -0.01, voxels=grid)  # (centres ON the surface land on either side of 0)
    assert inside.shape == (9 ** 3, 3) and float(inside.abs().max()) <= 0.08 + 1e-6  # the centres at multiples of 0.02 up to 0.08
    view = sdf.get_voxel_view(grid)
    q = torch.tensor([[0.0, 0.0, 0.0], [0.16, 0.0, 0.0], [0.5, 0.0, 0.0]], device="cuda")
    assert torch.allclose(view[q].cpu(), torch.tensor([-0.1, 0.06, 0.4]), atol=1e-6)  # the last one: outside the grid, asked of the mesh
    default = sdf.get_voxel_view(device="cuda")
    assert default.shape == (41, 41, 41)  # 0.01 m over the bounding box + 0.1
    cached = pv.CachedSDF("cube", 0.02, obj.bounding_box(padding=0.1), sdf, device="cuda", cache_path=None)
    assert cached.get_voxel_view() is cached.voxels
    other = cached.get_voxel_view(grid)
    assert other.shape == (21, 21, 21) and torch.equal(other.raw_data, view.raw_data)
    assert torch.allclose(other[q].cpu(), torch.tensor([-0.1, 0.06, 0.4]), atol=1e-6)


def test_numpy_input_and_dtype_device_round_trip():
    obj = factory("probe.obj")
    pts = H.uniform_points(100, obj.bounding_box(0.01)[:, 0], obj.bounding_box(0.01)[:, 1], seed=4)
    r_np = obj.object_frame_closest_point(pts.numpy())
    r_f64 = obj.object_frame_closest_point(pts.double())
    assert r_np.distance.dtype == torch.float32 and r_np.distance.device.type == "cpu"
    assert r_f64.distance.dtype == torch.float64
    assert torch.equal(r_np.distance.double(), r_f64.distance)


def test_sharded_index_base_reproduces_the_unsharded_jitter():
    obj = factory("probe.obj")
    pts = H.uniform_points(2000, obj.bounding_box(0.01)[:, 0], obj.bounding_box(0.01)[:, 1], seed=6).cuda()
    full = obj.object_frame_closest_point(pts)
    a = obj.object_frame_closest_point(pts[:777], index_base=0)
    b = obj.object_frame_closest_point(pts[777:], index_base=777)
    assert torch.equal(torch.cat((a.distance, b.distance)), full.distance)


def test_cached_sdf_built_from_the_mesh_kernel_is_within_one_voxel_of_ground_truth():
    """sdf.py:584-590 (debug_check_sdf): in-range cached values are within one resolution of the mesh SDF."""
    obj = factory("probe.obj")
    gt = pv.MeshSDF(obj)
    res = 0.002
    cached = pv.CachedSDF("probe", res, obj.bounding_box(padding=0.01), gt, device="cuda", cache_path=None)
    pts = H.uniform_points(20_000, obj.bounding_box(0.01)[:, 0], obj.bounding_box(0.01)[:, 1], seed=2).cuda()
    v, _ = cached(pts)
    vgt, _ = gt(pts)
    inb = cached.voxels.get_valid_values(pts)
    assert ((v - vgt).abs() < res)[inb].all()


def test_large_sphere_mesh_distance_close_to_analytic():
    m = mesh_io.uv_sphere_mesh(0.1, 96, 48)
    obj = pv.MeshObjectFactory(mesh=m)
    pts = H.uniform_points(4000, [-0.2] * 3, [0.2] * 3, seed=3)
    val, _ = pv.MeshSDF(obj)(pts.cuda())
    ref = pts.norm(dim=-1) - 0.1
    assert (val.cpu() - ref).abs().max() < 0.1 * (1 - np.cos(np.pi / 48)) + 1e-6
    assert ((val.cpu() < 0) == (ref < -1e-3))[ref.abs() > 1e-3].all()


@pytest.mark.parametrize("n,slices", [(300, 16), (70_000, 16), (300_000, 4)])
def test_every_slice_configuration_matches_oracle(n, slices):
    """The kernel picks 16, 8 or 4 waves per 64-point group from the point count and the mesh size (8: many points AND
    more than 128 tiles, covered by the sphere / chamfer tests); unsorted (n < 2048) and Morton-sorted processing
    orders are both covered.  All must reproduce the plain double loop bit for bit."""
    obj = factory("probe.obj")
    bb = obj.bounding_box(padding_ratio=0.5)
    pts = H.uniform_points(n, bb[:, 0], bb[:, 1], seed=n)
    assert_query_matches(obj, pts, seed=n)


def test_eight_wave_configuration_on_a_mesh_of_many_tiles():
    """Many points AND more than 128 tiles (> 32,768 triangles) -> 8 waves per point group.  The oracle checks a prefix
    of the batch (points are independent; the jitter is indexed by global point id)."""
    obj = pv.MeshObjectFactory(mesh=mesh_io.uv_sphere_mesh(0.1, 150, 120, scale=(1.0, 0.7, 1.3)))
    assert obj.num_faces > 128 * 256
    pts = H.uniform_points(1 << 18, [-0.16] * 3, [0.16] * 3, seed=8)
    obj.jitter_seed = 5
    res = obj.object_frame_closest_point(pts.cuda(), compute_normal=True)
    n = 12_000
    oc, od, og, of, on = oracle.mesh_query(H.oracle_mesh_from_factory(obj), pts[:n].numpy(), seed=5)
    assert np.array_equal(obj._last_face_ids[:n].cpu().numpy(), of)
    assert np.array_equal(res.closest[:n].cpu().numpy(), oc)
    assert np.array_equal(res.distance[:n].cpu().numpy(), od)
    assert np.array_equal(res.gradient[:n].cpu().numpy(), og)


def test_heavy_point_groups_are_handed_over_with_the_same_bits():
    """A sphere mesh of more than 128 tiles and points around its centre -- about equidistant to the whole surface: such
    point groups are handed over to a second launch that spreads their tiles over many workgroups (scratch given), or walk
    the mesh on their own 8 waves (scratch withheld).  Same bits either way, and the oracle's on a slice that contains
    centre points.  The chamfer entry point reports how many groups it handed over."""
    import ctypes
    from pytorch_volumetric_amd import _lib
    obj = pv.MeshObjectFactory(mesh=mesh_io.uv_sphere_mesh(0.1, 150, 120))
    assert obj.num_faces > 128 * 256
    g = torch.Generator().manual_seed(3)
    centre = (torch.rand(4096, 3, generator=g) - 0.5) * 0.02
    pts = torch.cat((centre[:700], H.uniform_points((1 << 17) - 4096, [-0.15] * 3, [0.15] * 3, seed=9), centre[700:])).float()
    obj.jitter_seed = 11
    outs = []
    for split in (True, False):
        obj.tile_split = split
        r = obj.object_frame_closest_point(pts.cuda(), compute_normal=True)
        outs.append((r.closest.clone(), r.distance.clone(), r.gradient.clone(), obj._last_face_ids.clone()))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    n = 1500
    oc, od, og, of, on = oracle.mesh_query(H.oracle_mesh_from_factory(obj), pts[:n].numpy(), seed=11)
    assert np.array_equal(outs[0][3][:n].cpu().numpy(), of) and np.array_equal(outs[0][0][:n].cpu().numpy(), oc)
    assert np.array_equal(outs[0][1][:n].cpu().numpy(), od) and np.array_equal(outs[0][2][:n].cpu().numpy(), og)
    # chamfer: with / without the scratch, and the number of groups listed
    lib = _lib.load()
    desc = obj._mesh_desc()
    dev_pts = pts.cuda().contiguous()
    W = torch.eye(4).unsqueeze(0).repeat(2, 1, 1).cuda().contiguous()
    W[1, :3, 3] = torch.tensor([0.01, 0.0, -0.02])
    order = _lib.morton_order(dev_pts)
    sums = torch.empty((2, 2), dtype=torch.float64, device="cuda")
    scratch = torch.zeros((_lib.mesh_scratch_bytes(pts.shape[0]) // 8,), dtype=torch.int64, device="cuda")
    for k, sc in enumerate((None, scratch)):
        _lib.check(lib.pvamd_chamfer_mesh(ctypes.byref(desc), _lib.ptr(W), 2, _lib.ptr(dev_pts), _lib.ptr(order), pts.shape[0],
                                          1000.0, _lib.ptr(sums[k]), _lib.ptr(sc), _lib.stream_ptr()), "pvamd_chamfer_mesh")
    listed = int(scratch.view(torch.int32)[0].item())
    assert 0 < listed < 2 * (pts.shape[0] // 64), listed
    assert torch.allclose(sums[0], sums[1], rtol=1e-12, atol=0)
    # more heavy groups than the list has slots for (8192): the ones that find it full walk the mesh themselves
    many = ((torch.rand(9000 * 64, 3, generator=g) - 0.5) * 0.02).float().cuda().contiguous()
    order = _lib.morton_order(many)
    scratch = torch.zeros((_lib.mesh_scratch_bytes(many.shape[0]) // 8,), dtype=torch.int64, device="cuda")
    one = torch.empty((2, 1), dtype=torch.float64, device="cuda")
    for k, sc in enumerate((None, scratch)):
        _lib.check(lib.pvamd_chamfer_mesh(ctypes.byref(desc), _lib.ptr(W), 1, _lib.ptr(many), _lib.ptr(order), many.shape[0],
                                          1000.0, _lib.ptr(one[k]), _lib.ptr(sc), _lib.stream_ptr()), "pvamd_chamfer_mesh")
    assert int(scratch.view(torch.int32)[0].item()) > _lib.mesh_scratch_slots(many.shape[0]) == _lib.MESH_SCRATCH_GROUPS
    assert torch.allclose(one[0], one[1], rtol=1e-12, atol=0)
    obj.tile_split = True
    a_ = obj.object_frame_closest_point(many[:140_000]).distance.clone()
    obj.tile_split = False
    assert torch.equal(a_, obj.object_frame_closest_point(many[:140_000]).distance)


def test_culling_is_exact_for_far_and_degenerate_queries():
    """Points far from the mesh (every tile 'far'), on vertices / edges (distance 0, ties), and NaN."""
    obj = factory("box_template.obj")
    far = H.uniform_points(3000, [50.0] * 3, [60.0] * 3, seed=1)
    corners = torch.tensor([[1.0, 1.0, 1.0], [1.0, 0.0, 1.0], [0.0, 0.0, 1.0], [-1.0, 1.0, -1.0], [0.3, -1.0, 0.2]])
    pts = torch.cat((far, corners.repeat(200, 1), H.uniform_points(2000, [-1.001] * 3, [1.001] * 3, seed=2)))
    assert_query_matches(obj, pts, seed=11)
    nan_pt = torch.tensor([[float("nan"), 0.0, 0.0], [0.5, 0.5, 3.0]])
    res = obj.object_frame_closest_point(nan_pt.cuda())
    assert torch.isnan(res.distance[0]) and abs(res.distance[1].item() - 2.0) < 1e-6


def test_reference_debug_assertions_hold_for_a_mesh_backed_cache():
    """The reference's in-source checks (sdf.py:508-512, 574-590, enabled by debug_check_sdf): voxel centres read back
    their own value; the BOUNDING_BOX fallback under-approximates the true distance and points roughly the same way;
    in-range values are within one resolution of ground truth."""
    obj = factory("ycb_power_drill.npz")
    gt = pv.MeshSDF(obj)
    cached = pv.CachedSDF("drill", 0.01, obj.bounding_box(padding=0.1), gt, device="cuda", cache_path=None,
                          debug_check_sdf=True)  # runs the centre check in __init__ and the one-voxel check per call
    lo = np.array([r[0] for r in cached.ranges]) - 0.3
    hi = np.array([r[1] for r in cached.ranges]) + 0.3
    pts = H.uniform_points(20_000, lo, hi, seed=3).cuda()
    val, grad = cached(pts)
    oob = ~cached.voxels.get_valid_values(pts)
    v_gt, g_gt = gt(pts[oob])
    assert (v_gt - val[oob] > 0).all()                                   # sdf.py:576-578
    cos = torch.cosine_similarity(g_gt, grad[oob], dim=-1)               # sdf.py:580-582
    assert (cos > 0.7).all() and cos.mean() > 0.95


def nasty_mesh(rng, offset):
    """Triangles the broad-phase bounds must stay conservative for: needle slivers, zero-area (collinear and repeated
    corners), duplicates, a few huge faces next to tiny ones, all far from the origin (large |coordinate| / size)."""
    tris = []
    for _ in range(150):  # needles: two corners ~1e-6 apart, third far along a random direction
        a = rng.uniform(-1, 1, 3)
        d = rng.normal(size=3)
        tris.append([a, a + 1e-6 * rng.normal(size=3), a + d / np.linalg.norm(d) * rng.uniform(0.5, 3.0)])
    for _ in range(60):  # collinear
        a, d = rng.uniform(-1, 1, 3), rng.normal(size=3)
        tris.append([a, a + 0.3 * d, a + 0.7 * d])
    for _ in range(30):  # a point, three times
        a = rng.uniform(-1, 1, 3)
        tris.append([a, a, a])
    for _ in range(200):  # tiny
        a = rng.uniform(-1, 1, 3)
        tris.append([a, a + 1e-3 * rng.normal(size=3), a + 1e-3 * rng.normal(size=3)])
    for _ in range(8):  # huge
        tris.append(list(rng.uniform(-4, 4, (3, 3))))
    tris = tris + tris[:40]  # exact duplicates: ties must resolve to the lowest face id
    soup = np.asarray(tris, dtype=np.float64) + np.asarray(offset)
    verts = soup.reshape(-1, 3)
    return mesh_io.TriMesh(verts, np.arange(len(verts)).reshape(-1, 3))


@pytest.mark.parametrize("offset,scale", [((0.0, 0.0, 0.0), 1.0), ((300.0, -200.0, 100.0), 1.0),
                                          ((0.0, 0.0, 0.0), 1e-3), ((2.0, 1.0, -3.0), 1e3), ((5e3, 5e3, 5e3), 1.0)])
def test_slivers_degenerate_and_duplicate_triangles_match_oracle(offset, scale):
    """also at mm and km scale, and 5 km from the origin (fp32 cancellation in every bound)"""
    rng = np.random.default_rng(7)
    obj = pv.MeshObjectFactory(mesh=nasty_mesh(rng, offset).scaled(scale))
    bb = obj.bounding_box(padding_ratio=0.1)
    pts = H.uniform_points(6000, bb[:, 0], bb[:, 1], seed=5)
    # plus points exactly on corners, straddling the needles, and in the planes of the huge faces
    on = torch.from_numpy(obj._mesh.vertices[::7].astype(np.float32))
    soup = obj._mesh.triangle_soup()[-48:-40].astype(np.float32)  # the huge faces
    w = np.random.default_rng(11).dirichlet((0.3, 0.3, 0.3), size=(8, 60)).astype(np.float32)
    in_plane = torch.from_numpy(np.einsum("fsk,fkd->fsd", w * 1.6 - 0.2, soup).reshape(-1, 3))
    assert_query_matches(obj, torch.cat((pts, on, on + 1e-4 * float(scale), in_plane)), seed=3)


def test_points_aabb_ignores_non_finite_coordinates():
    from pytorch_volumetric_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(100_000, 3)).astype(np.float32) * np.array([1.0, 50.0, 1e-3], dtype=np.float32)
    pts[5] = [np.nan, np.inf, -np.inf]
    pts[77, 1] = np.nan
    d = torch.from_numpy(pts).cuda()
    box = torch.empty((2, 3), dtype=torch.float32, device="cuda")
    _lib.check(lib.pvamd_points_aabb(_lib.ptr(d), d.shape[0], _lib.ptr(box), _lib.stream_ptr()), "aabb")
    fin = np.where(np.isfinite(pts), pts, np.nan)
    assert np.array_equal(box.cpu().numpy(), np.stack((np.nanmin(fin, axis=0), np.nanmax(fin, axis=0))))
    _lib.check(lib.pvamd_points_aabb(None, 0, _lib.ptr(box), _lib.stream_ptr()), "aabb empty")
    assert np.array_equal(box.cpu().numpy(), np.array([[np.inf] * 3, [-np.inf] * 3], dtype=np.float32))


class _ViaGenericPath(pv.MeshSDF):
    """A MeshSDF that CachedSDF does not recognise as one: the cache is then built through gt_sdf(points) + pvamd_pack_grid."""


@pytest.mark.parametrize("mesh,res,pad", [("ycb_power_drill.npz", 0.01, 0.1), ("ycb_power_drill.npz", 0.02, 0.013),
                                          ("box_template.obj", 0.25, 0.3), ("offset_wrench_nogrip.obj", 0.004, 0.05)])
def test_fused_cache_build_writes_the_bits_of_the_generic_build(mesh, res, pad, tmp_path):
    """Round 6: pvamd_cache_build -- voxel centres + a 4 x 4 x 4-brick processing order in one launch, the
    mesh kernel writing packed records -- against the generic construction (cartesian_prod, Hilbert sort, mesh query, pack):
    the same cache, bit for bit, for grids whose sides are and are not multiples of 4; and the reference-format pickle."""
    obj = factory(mesh)
    rng = obj.bounding_box(padding=pad)
    path = str(tmp_path / "sdf_cache.pkl")
    fused = pv.CachedSDF(mesh, res, rng, pv.MeshSDF(obj), device="cuda", cache_path=path)
    generic = pv.CachedSDF(mesh, res, rng, _ViaGenericPath(obj), device="cuda", cache_path=None)
    assert fused._packed.shape == generic._packed.shape and fused._view.shape == generic._view.shape
    assert torch.equal(fused._packed.view(torch.int32), generic._packed.view(torch.int32))
    val, grad = torch.load(path, weights_only=False)[fused.name]
    assert val.shape == tuple(fused._view.shape) and grad.shape == (fused._packed.shape[0], 3) and val.is_contiguous()
    assert torch.equal(val.reshape(-1), generic._packed[:, 0].cpu()) and torch.equal(grad, generic._packed[:, 1:4].cpu())
    again = pv.CachedSDF(mesh, res, rng, pv.MeshSDF(obj), device="cuda", cache_path=path)  # from the pickle: nothing is rebuilt
    assert torch.equal(again._packed.view(torch.int32), fused._packed.view(torch.int32))
    # and a slice of it against the oracle's mesh query at the same global indices
    _, pts = pv.get_coordinates_and_points_in_grid(res, fused.ranges)
    n = len(pts)
    a = n // 3
    b = min(n, a + 300)
    oc, od, og, of, on = oracle.mesh_query(H.oracle_mesh_from_factory(obj), pts[a:b].numpy(), seed=obj.jitter_seed, index_base=a)
    assert np.array_equal(fused._packed[a:b, 0].cpu().numpy(), od, equal_nan=True)
    assert np.array_equal(fused._packed[a:b, 1:4].cpu().numpy(), og, equal_nan=True)
