"""-m gpu: RobotSDF (FK -> MFMA transform stack -> fused composed query), BASELINE config C4 shape."""
import os

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from oracle import oracle
from pytorch_volumetric_amd import mesh_io
from pytorch_volumetric_amd import transforms as tf
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_transform_stack_mfma_matches_oracle_bitwise():
    lib = pv._lib.load()
    S, A = 8, 37  # not a multiple of the 16 matrices one wave handles
    offs = H.random_rigid(S, seed=1, trans=0.05)
    world = H.random_rigid(S * A, seed=2, trans=1.0)
    off_inv = tf.rigid_inverse(offs).cuda().contiguous()
    world_d = world.cuda().contiguous()
    out = torch.empty_like(world_d)
    pv._lib.check(lib.pvamd_transform_stack(pv._lib.ptr(off_inv), pv._lib.ptr(world_d), S, A, pv._lib.ptr(out),
                                            pv._lib.stream_ptr()), "pvamd_transform_stack")
    expect = oracle.transform_stack(off_inv.cpu().numpy(), world.numpy(), S, A)
    assert np.array_equal(out.cpu().numpy(), expect)
    # and it is what the reference computes: offset^-1 @ world^-1
    ref = (tf.rigid_inverse(offs).double()[:, None] @ tf.rigid_inverse(world).double().reshape(S, A, 4, 4)).reshape(-1, 4, 4)
    assert np.allclose(out.cpu().double().numpy(), ref.numpy(), atol=1e-6)


def synthetic_arm(tmp_path, n_links=8):
    """KUKA-like 7-revolute serial chain with 8 mesh links (the KUKA assets are not available offline)."""
    names = []
    for i in range(n_links):
        m = mesh_io.uv_sphere_mesh(1.0, 16, 8, scale=(0.05, 0.05, 0.09), center=(0, 0, 0.08))
        p = os.path.join(tmp_path, f"link_{i}.obj")
        mesh_io.save_obj(p, m)
        names.append(f"link_{i}.obj")
    axes = ["0 0 1", "0 1 0", "0 0 1", "0 -1 0", "0 0 1", "0 1 0", "0 0 1"]
    parts = ['<robot name="arm7">']
    for i in range(n_links):
        parts.append(f'<link name="link_{i}"><visual><origin xyz="0 0 0.01" rpy="0 0 0.1"/><geometry>'
                     f'<mesh filename="{names[i]}"/></geometry></visual></link>')
    for i in range(n_links - 1):
        parts.append(f'<joint name="j{i}" type="revolute"><parent link="link_{i}"/><child link="link_{i + 1}"/>'
                     f'<origin xyz="0 0 0.18" rpy="0 0 0"/><axis xyz="{axes[i % 7]}"/></joint>')
    parts.append('</robot>')
    return pv.build_serial_chain_from_urdf("\n".join(parts), f"link_{n_links - 1}")


def test_robot_sdf_config_batch_equals_loop_and_shapes(tmp_path):
    """tests/test_model_to_sdf.py:173-212, 301-326 of the reference, headless."""
    chain = synthetic_arm(str(tmp_path))
    s = pv.RobotSDF(chain, path_prefix=str(tmp_path),
                    link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=0.2, device="cuda",
                                                           cache_path=None))
    assert len(s.sdf.sdfs) == 8 and len(s.joint_names) == 7
    th0 = torch.tensor([0.0, -np.pi / 4, 0.0, np.pi / 2, 0.0, np.pi / 4, 0.0])
    N = 20
    g = torch.Generator().manual_seed(0)
    th = torch.cat((th0.view(1, -1), th0 + torch.randn(N - 1, 7, generator=g) * 0.1))
    _, pts = pv.get_coordinates_and_points_in_grid(0.02, [(-0.6, 0.6), (0.02, 0.02), (-0.2, 1.4)])
    pts = pts.cuda()
    s.set_joint_configuration(th)
    all_val, all_grad = s(pts)
    assert all_val.shape == (N, len(pts)) and all_grad.shape == (N, len(pts), 3)
    for i in range(N):
        s.set_joint_configuration(th[i])
        v, gr = s(pts)
        assert v.shape == (len(pts),)
        assert torch.allclose(v, all_val[i])
        assert torch.allclose(gr.nan_to_num(0.), all_grad[i].nan_to_num(0.), atol=1e-6)
    # multi-dim configuration and point batches (reference :301-326)
    s.set_joint_configuration(th.view(4, 5, 7))
    bv, bg = s(pts[:600].view(2, 300, 3))
    assert bv.shape == (4, 5, 2, 300) and bg.shape == (4, 5, 2, 300, 3)
    assert torch.equal(bv.reshape(N, -1), all_val[:, :600])
    bbs = s.link_bounding_boxes()
    assert bbs.shape == (8, 4, 5, 8, 3) or bbs.shape[0] == 8
    # some points are inside the arm, most outside
    assert (all_val < 0).any() and (all_val > 0.1).any()


def test_transformed_link_meshes_lie_on_the_robot_surface(tmp_path):
    """visualization.get_transformed_meshes (visualization.py:83-107 of the reference): every link's mesh placed by the
    current configuration -- its vertices are surface points of the robot SDF (never outside it), for MeshSDF links and,
    through gt_sdf, for cached ones; `obj_to_world_tsf` composes on the left."""
    chain = synthetic_arm(str(tmp_path), n_links=4)
    q = torch.tensor([0.3, -0.6, 0.2])
    for cls in (pv.MeshSDF, pv.cache_link_sdf_factory(resolution=0.01, padding=0.05, device="cuda", cache_path=None)):
        s = pv.RobotSDF(chain, path_prefix=str(tmp_path), link_sdf_cls=cls)
        s.set_joint_configuration(q)
        meshes = pv.get_transformed_meshes(s)
        assert len(meshes) == 4 and all(isinstance(m, mesh_io.TriMesh) for m in meshes)
        exact = pv.RobotSDF(chain, path_prefix=str(tmp_path))
        exact.set_joint_configuration(q)
        for m in meshes:
            v, _ = exact(torch.tensor(m.vertices, dtype=torch.float32).cuda())
            assert float(v.max()) < 1e-5  # on this link's surface (or inside a neighbouring link)
        assert float(np.abs(np.stack([m.vertices.mean(axis=0) for m in meshes])[:, 2]).max()) > 0.2  # links were placed
        shift = pv.Transform3d(pos=torch.tensor([[1.0, -2.0, 0.5]]))
        moved = pv.get_transformed_meshes(s, obj_to_world_tsf=shift)
        for m, n in zip(meshes, moved):
            assert np.allclose(n.vertices, m.vertices + np.array([1.0, -2.0, 0.5]), atol=1e-6) and np.array_equal(n.faces, m.faces)
    s.set_joint_configuration(torch.zeros(5, 3))
    with pytest.raises(ValueError):
        pv.get_transformed_meshes(s)  # one mesh set per configuration: a batch is refused


def test_robot_sdf_wrench_urdf_single_mesh_link():
    """The reference's own URDF fixture (tests/offset_wrench.urdf): 6-DOF chain, one mesh link."""
    chain = pv.build_chain_from_urdf(open(H.mesh_path("offset_wrench.urdf")).read())
    s = pv.RobotSDF(chain, path_prefix=H.MESHES,
                    link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.005, padding=0.05, device="cuda",
                                                           cache_path=None))
    assert len(s.sdf.sdfs) == 1 and len(s.joint_names) == 6
    q = torch.tensor([0.1, -0.05, 0.2, 0.3, -0.2, 0.5])
    s.set_joint_configuration(q)
    # a point given in the wrench's own frame, mapped to the world by FK, must see the leaf's own value
    leaf = s.sdf.sdfs[0]
    local = H.uniform_points(500, leaf.surface_bounding_box()[:, 0], leaf.surface_bounding_box()[:, 1], seed=1)
    fk = chain.forward_kinematics(q)["offset_wrench"].get_matrix()[0]
    world = local @ fk[:3, :3].T + fk[:3, 3]
    v_world, _ = s(world.cuda())
    v_local, _ = leaf(local.cuda())
    # fp32 round trip through FK^-1 moves points by ~1e-7: allow the few that cross a voxel boundary
    assert (torch.isclose(v_world, v_local, atol=1e-6).float().mean() > 0.97)


def test_on_device_fk_matches_oracle_bitwise_and_torch_fk(tmp_path):
    chain = synthetic_arm(str(tmp_path))
    leaves = [f"link_{i}" for i in range(8)]
    raw = chain.joint_table(leaves)
    import ctypes
    F = len(raw) // ctypes.sizeof(pv._lib.JointDesc)
    A, M = 77, 7
    q = torch.randn(A, M, generator=torch.Generator().manual_seed(1)) * 0.8
    qd = q.cuda().contiguous()
    sq, cq = torch.sin(qd), torch.cos(qd)
    joints = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    scratch = torch.empty((F, 12, A), device="cuda")
    lw = torch.empty((8 * A, 4, 4), device="cuda")
    lib = pv._lib.load()
    pv._lib.check(lib.pvamd_chain_fk(pv._lib.ptr(joints), F, pv._lib.ptr(qd), pv._lib.ptr(sq), pv._lib.ptr(cq), A, M,
                                     pv._lib.ptr(scratch), pv._lib.ptr(lw), pv._lib.stream_ptr()), "pvamd_chain_fk")
    # the oracle with the SAME sin/cos arrays (device sinf/cosf differ from libm in the last ulp)
    joints_o = (oracle.OracleJoint * F).from_buffer_copy(raw)
    world = np.zeros((F, A, 12), np.float32)
    olw = np.zeros((8 * A, 4, 4), np.float32)
    import ctypes as C
    qn, sn, cn = q.numpy(), sq.cpu().numpy(), cq.cpu().numpy()
    oracle.load().oracle_chain_fk(joints_o, C.c_int32(F), C.c_void_p(qn.ctypes.data), C.c_void_p(sn.ctypes.data),
                                  C.c_void_p(cn.ctypes.data), C.c_int32(A), C.c_int32(M), C.c_void_p(world.ctypes.data),
                                  C.c_void_p(olw.ctypes.data))
    assert np.array_equal(lw.cpu().numpy(), olw)
    assert np.array_equal(scratch.cpu().numpy().transpose(0, 2, 1), world)
    fk = chain.forward_kinematics(q)
    ref = torch.cat([fk[n].get_matrix() for n in leaves])
    assert torch.allclose(lw.cpu(), ref, atol=2e-6)


@pytest.mark.parametrize("A", [1, 20, 77, 200])
def test_one_launch_configure_matches_the_oracle_given_the_sin_cos_it_used(tmp_path, A):
    """pvamd_configure_chain = sin / cos + the frame walk + offset^-1 o world^-1 (MFMA) in one kernel.  It emits the sines and
    cosines it used; the oracle's FK + transform stack fed the same numbers must give the same bits (model_to_sdf.py:94-113)."""
    import ctypes as C
    chain = synthetic_arm(str(tmp_path))
    leaves = [f"link_{i}" for i in range(8)]
    raw = chain.joint_table(leaves)
    F = len(raw) // C.sizeof(pv._lib.JointDesc)
    S, M = 8, 7
    q = torch.randn(A, M, generator=torch.Generator().manual_seed(A)) * 0.8
    qd = q.cuda().contiguous()
    joints = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    off = H.random_rigid(S, seed=5, trans=0.1)
    off_inv = tf.rigid_inverse(off).cuda().contiguous()
    sincos = torch.full((A, M, 2), 9.0, device="cuda")
    scratch = torch.empty((F, 12, A), device="cuda")
    lw = torch.empty((S * A, 4, 4), device="cuda")
    stack = torch.empty((S * A, 4, 4), device="cuda")
    lib = pv._lib.load()
    pv._lib.check(lib.pvamd_configure_chain(pv._lib.ptr(joints), F, pv._lib.ptr(qd), A, M, pv._lib.ptr(off_inv), S,
                                            pv._lib.ptr(sincos), pv._lib.ptr(scratch), pv._lib.ptr(lw), pv._lib.ptr(stack),
                                            pv._lib.stream_ptr()), "pvamd_configure_chain")
    sc = sincos.cpu().numpy()
    sn, cn = np.ascontiguousarray(sc[..., 0]), np.ascontiguousarray(sc[..., 1])
    assert np.allclose(sn, np.sin(q.numpy()), atol=1e-6) and np.allclose(cn, np.cos(q.numpy()), atol=1e-6)
    joints_o = (oracle.OracleJoint * F).from_buffer_copy(raw)
    world = np.zeros((F, A, 12), np.float32)
    olw = np.zeros((S * A, 4, 4), np.float32)
    qn = q.numpy()
    oracle.load().oracle_chain_fk(joints_o, C.c_int32(F), C.c_void_p(qn.ctypes.data), C.c_void_p(sn.ctypes.data),
                                  C.c_void_p(cn.ctypes.data), C.c_int32(A), C.c_int32(M), C.c_void_p(world.ctypes.data),
                                  C.c_void_p(olw.ctypes.data))
    assert np.array_equal(lw.cpu().numpy(), olw)
    assert np.array_equal(stack.cpu().numpy(), oracle.transform_stack(off_inv.cpu().numpy(), olw, S, A))
    # the two-kernel path of round 3 on the same sines / cosines: the same stack
    lw2 = torch.empty_like(lw)
    st2 = torch.empty_like(stack)
    sin_d, cos_d = sincos[..., 0].contiguous(), sincos[..., 1].contiguous()  # kept alive until the kernels have run
    pv._lib.check(lib.pvamd_chain_fk(pv._lib.ptr(joints), F, pv._lib.ptr(qd), pv._lib.ptr(sin_d), pv._lib.ptr(cos_d), A, M,
                                     pv._lib.ptr(scratch), pv._lib.ptr(lw2), pv._lib.stream_ptr()), "pvamd_chain_fk")
    pv._lib.check(lib.pvamd_transform_stack(pv._lib.ptr(off_inv), pv._lib.ptr(lw2), S, A, pv._lib.ptr(st2), pv._lib.stream_ptr()),
                  "pvamd_transform_stack")
    torch.cuda.synchronize()
    assert torch.equal(lw2, lw) and torch.equal(st2, stack)


def test_configure_and_query_into_is_two_launches_and_graph_capturable(tmp_path):
    """robot.configure_and_query_into(q, pts, val, grad) with q on the GPU == set_joint_configuration(q) + robot(pts), bit for
    bit, from host tensors or device tensors, eagerly or replayed from a hipGraph with new joint values in place."""
    robot = pv.RobotSDF(synthetic_arm(str(tmp_path)), path_prefix=str(tmp_path),
                        link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=0.1, device="cuda", cache_path=None))
    A, P = 24, 5000
    g = torch.Generator().manual_seed(3)
    q1, q2 = torch.randn(A, 7, generator=g) * 0.6, torch.randn(A, 7, generator=g) * 0.6
    pts = H.uniform_points(P, [-0.5, -0.5, -0.1], [0.5, 0.5, 1.2], seed=2).cuda()
    robot.set_joint_configuration(q1)  # host tensor: one H2D + one launch
    v_host, g_host = robot(pts)
    robot.set_joint_configuration(q1.cuda())  # device tensor: one launch
    v_dev, g_dev = robot(pts)
    assert torch.equal(v_host, v_dev) and torch.equal(g_host.nan_to_num(7.0), g_dev.nan_to_num(7.0))
    val = torch.empty((A, P), device="cuda")
    grad = torch.empty((A, P, 3), device="cuda")
    qd = q1.cuda().contiguous()
    robot.configure_and_query_into(qd, pts, val, grad)
    assert torch.equal(val, v_host) and torch.equal(grad.nan_to_num(7.0), g_host.nan_to_num(7.0))
    assert robot.configuration_batch == (A,)
    bb_fused = robot.sdf.surface_bounding_box(padding=0.0).clone()
    robot.set_joint_configuration(q1)
    assert torch.equal(bb_fused, robot.sdf.surface_bounding_box(padding=0.0))
    # captured once, replayed with other joint values written into the same device tensor
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        robot.configure_and_query_into(qd, pts, val, grad)
        with torch.cuda.graph(graph, stream=side):
            robot.configure_and_query_into(qd, pts, val, grad)
    torch.cuda.current_stream().wait_stream(side)
    qd.copy_(q2)
    graph.replay()
    torch.cuda.synchronize()
    robot.set_joint_configuration(q2)
    v2, g2 = robot(pts)
    assert torch.equal(val, v2) and torch.equal(grad.nan_to_num(7.0), g2.nan_to_num(7.0))


def test_float64_queries_follow_a_stack_rewritten_in_place(tmp_path):
    """configure_and_query_into re-writes its transform stack in place and skips set_transforms from the second call
    on; the float64 widening of that stack (the float64 query path's own copy) must not survive it."""
    robot = pv.RobotSDF(synthetic_arm(str(tmp_path)), path_prefix=str(tmp_path),
                        link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=0.1, device="cuda", cache_path=None))
    A, P = 6, 3000
    g = torch.Generator().manual_seed(5)
    q1, q2 = (torch.randn(A, 7, generator=g) * 0.6).cuda(), (torch.randn(A, 7, generator=g) * 0.6).cuda()
    pts = H.uniform_points(P, [-0.5, -0.5, -0.1], [0.5, 0.5, 1.2], seed=4).cuda()
    val, grad = torch.empty((A, P), device="cuda"), torch.empty((A, P, 3), device="cuda")
    robot.configure_and_query_into(q1, pts, val, grad)
    v1, _ = robot(pts.double())
    frames1 = robot.object_to_link_frames.get_matrix().clone()
    held = robot.object_to_link_frames  # what a caller keeps must not change under it on the next call
    held_composed = robot.sdf.obj_frame_to_link_frame
    robot.configure_and_query_into(q2, pts, val, grad)
    v2, g2 = robot(pts.double())
    assert torch.equal(held.get_matrix(), frames1) and torch.equal(held_composed.get_matrix(), frames1)
    assert not torch.equal(robot.object_to_link_frames.get_matrix(), frames1)
    bb2 = robot.sdf.surface_bounding_box(padding=0.0).clone()
    robot.set_joint_configuration(q2)
    w2, h2 = robot(pts.double())
    assert v2.dtype == torch.float64 and not torch.equal(v1, v2)
    assert torch.equal(v2, w2) and torch.equal(g2.nan_to_num(7.0), h2.nan_to_num(7.0))
    assert torch.equal(bb2, robot.sdf.surface_bounding_box(padding=0.0))


@pytest.mark.parametrize("n_links", [30, 60])
def test_robots_with_many_links_configure(tmp_path, n_links):
    """the one-launch configure kernel stages the leaf frames of 64 configurations in LDS -- more than 64 KB of it
    from 21 SDF-carrying links on (an opt-in attribute of the launch), and past ~50 links it does not fit at all: those robots
    take sin / cos + pvamd_chain_fk + pvamd_transform_stack.  Both against the oracle's FK + stack, bit for bit."""
    chain = synthetic_arm(str(tmp_path), n_links=n_links)
    robot = pv.RobotSDF(chain, path_prefix=str(tmp_path),
                        link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.04, padding=0.1, device="cuda", cache_path=None))
    S, M, A = n_links, n_links - 1, 70
    assert len(robot.sdf.sdfs) == S and len(robot.joint_names) == M
    q = (torch.randn(A, M, generator=torch.Generator().manual_seed(n_links)) * 0.4).cuda()
    robot.set_joint_configuration(q)
    stack = robot.object_to_link_frames.get_matrix()
    assert stack.shape == (S * A, 4, 4) and bool(torch.isfinite(stack).all())
    # the same fma chains as the two-launch path, whichever the robot took
    lib = pv._lib.load()
    joints, F = robot._joint_table_dev(q.device)
    scratch = torch.empty((F, 12, A), device="cuda")
    lw = torch.empty((S * A, 4, 4), device="cuda")
    st = torch.empty((S * A, 4, 4), device="cuda")
    if n_links <= 40:
        sincos = torch.empty((A, M, 2), device="cuda")
        pv._lib.check(lib.pvamd_configure_chain(joints.data_ptr(), F, q.data_ptr(), A, M, robot._offset_inv_dev(q.device).data_ptr(), S,
                                                sincos.data_ptr(), scratch.data_ptr(), None, st.data_ptr(), pv._lib.stream_ptr()),
                      "pvamd_configure_chain")
        assert torch.equal(st, stack)
        sin_q, cos_q = sincos[..., 0].contiguous(), sincos[..., 1].contiguous()
    else:
        rc = lib.pvamd_configure_chain(joints.data_ptr(), F, q.data_ptr(), A, M, robot._offset_inv_dev(q.device).data_ptr(), S,
                                       None, scratch.data_ptr(), None, st.data_ptr(), pv._lib.stream_ptr())
        assert rc == pv._lib.E_SHAPE
        sin_q, cos_q = torch.sin(q), torch.cos(q)
        val, grad = torch.empty((A, 64), device="cuda"), torch.empty((A, 64, 3), device="cuda")
        with pytest.raises(ValueError, match="one-launch"):
            robot.configure_and_query_into(q, torch.zeros(64, 3, device="cuda"), val, grad)
    pv._lib.check(lib.pvamd_chain_fk(joints.data_ptr(), F, q.data_ptr(), sin_q.data_ptr(), cos_q.data_ptr(), A, M,
                                     scratch.data_ptr(), lw.data_ptr(), pv._lib.stream_ptr()), "pvamd_chain_fk")
    pv._lib.check(lib.pvamd_transform_stack(robot._offset_inv_dev(q.device).data_ptr(), lw.data_ptr(), S, A, st.data_ptr(),
                                            pv._lib.stream_ptr()), "pvamd_transform_stack")
    torch.cuda.synchronize()
    assert torch.equal(st, stack)
    # ... and the query over that many leaves answers like the per-configuration loop
    pts = H.uniform_points(2000, [-0.5, -0.5, -0.1], [0.5, 0.5, 0.18 * n_links], seed=9).cuda()
    v, g = robot(pts)
    robot.set_joint_configuration(q[3])
    v3, g3 = robot(pts)
    assert torch.equal(v[3], v3) and torch.equal(g[3].nan_to_num(7.0), g3.nan_to_num(7.0))


def test_joint_values_of_the_wrong_length_are_refused(tmp_path):
    """a float32 GPU vector used where it is must still hold M values (the kernel indexes q[a * M + joint])."""
    robot = pv.RobotSDF(synthetic_arm(str(tmp_path)), path_prefix=str(tmp_path),
                        link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.04, padding=0.1, device="cuda", cache_path=None))
    for bad in (torch.zeros(5, device="cuda"), torch.zeros(5), torch.zeros(3, 6, device="cuda"), torch.zeros(9)):
        with pytest.raises(ValueError, match="joint"):
            robot.set_joint_configuration(bad)
    robot.set_joint_configuration(torch.zeros(7, device="cuda"))


def test_query_into_and_graph_replay(tmp_path):
    """A planner-style inner loop: set_joint_configuration + query_into captured once in a hipGraph and replayed."""
    chain = synthetic_arm(str(tmp_path))
    s = pv.RobotSDF(chain, path_prefix=str(tmp_path),
                    link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=0.2, device="cuda", cache_path=None))
    A, P = 16, 4096
    q = torch.randn(A, 7, generator=torch.Generator().manual_seed(3)) * 0.5
    s.set_joint_configuration(q)
    pts = H.uniform_points(P, [-0.6, -0.6, -0.2], [0.6, 0.6, 1.4], seed=2).cuda().contiguous()
    ref_v, ref_g = s(pts)
    val = torch.empty((A, P), device="cuda")
    grad = torch.empty((A, P, 3), device="cuda")
    s.query_into(pts, val, grad)
    assert torch.equal(val, ref_v) and torch.equal(grad.nan_to_num(5.), ref_g.nan_to_num(5.))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            s.query_into(pts, val, grad)
    torch.cuda.current_stream().wait_stream(side)
    val.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(val, ref_v)


def test_sharded_sdf_over_rccl_single_rank():
    """The nccl (= RCCL) code path of ShardedSDF on one GPU: device tensors through all_gather_into_tensor."""
    import torch.distributed as dist
    from tests.test_cached_gpu import make_cached, query_points
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = make_cached()
        pts = query_points(c, 10_001, seed=4).cuda()
        v, g = c(pts)
        sv, sg = pv.ShardedSDF(c)(pts.reshape(73, 137, 3))
        assert sv.shape == (73, 137) and sv.is_cuda
        assert torch.equal(sv.reshape(-1), v) and torch.equal(sg.reshape(-1, 3).nan_to_num(3.), g.nan_to_num(3.))
        # a composition of cached grids goes out as packed records through ONE collective and an unpack kernel
        scene = pv.ComposedSDF([c, c, c], pv.Translate(0.05, 0, 0).stack(pv.Translate(-0.1, 0, 0.1), pv.Translate(0, 0.2, 0)))
        for batch in (None, 5):
            if batch is not None:
                t = torch.randn(3 * batch, 3, generator=torch.Generator().manual_seed(1)) * 0.05
                scene.set_transforms(pv.Translate(t), batch_dim=(batch,))
            dv, dg = scene(pts)
            sh = pv.ShardedSDF(scene)
            gv, gg = sh(pts)
            assert sh.last_path == "packed" and gv.shape == dv.shape and gg.shape == dg.shape
            assert torch.equal(gv, dv) and torch.equal(gg.nan_to_num(3.), dg.nan_to_num(3.))
            gv2, gg2 = sh(pts.reshape(73, 137, 3))
            assert torch.equal(gv2.reshape(dv.shape), dv) and torch.equal(gg2.reshape(dg.shape).nan_to_num(3.), dg.nan_to_num(3.))
    finally:
        dist.destroy_process_group()


def test_full_size_c4_properties(tmp_path):
    """BASELINE C4 at full size (8 links, 200 configurations x 262,144 points = 52.4M pairs, 839 MB of output): parity
    through size-independent properties -- config batch == single-config queries, determinism, a slice vs the oracle."""
    chain = synthetic_arm(str(tmp_path))
    s = pv.RobotSDF(chain, path_prefix=str(tmp_path),
                    link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=0.1, device="cuda", cache_path=None))
    A, P = 200, 1 << 18
    th0 = torch.tensor([0.0, -np.pi / 4, 0.0, np.pi / 2, 0.0, np.pi / 4, 0.0])
    th = torch.cat((th0.view(1, -1), th0 + torch.randn(A - 1, 7, generator=torch.Generator().manual_seed(0)) * 0.1))
    pts = H.uniform_points(P, [-0.7, -0.7, -0.2], [0.7, 0.7, 1.5], seed=1).cuda()
    s.set_joint_configuration(th)
    val, grad = s(pts)
    assert val.shape == (A, P) and grad.shape == (A, P, 3)
    stack = s.object_to_link_frames.get_matrix().clone()  # (S*A,4,4) leaf-major, from the FK + MFMA kernels
    v2, _ = s(pts)
    assert torch.equal(val, v2)
    for a in (0, 57, 199):
        s.set_joint_configuration(th[a])
        v, g = s(pts)
        assert torch.equal(v, val[a]) and torch.equal(g.nan_to_num(4.), grad[a].nan_to_num(4.))
    # oracle on a slice: 3 configurations x 20,000 points, same transform stack
    S = len(s.sdf.sdfs)
    sel = [0, 57, 199]
    tf_sel = stack.reshape(S, A, 4, 4)[:, sel].reshape(-1, 4, 4).cpu().numpy()
    ogrids = [H.oracle_grid_from_cached(leaf) for leaf in s.sdf.sdfs]
    ov, og, _ = oracle.composed_query(ogrids, tf_sel, len(sel), pts[:20_000].cpu().numpy())
    assert np.array_equal(val[sel, :20_000].cpu().numpy(), ov, equal_nan=True)
    assert np.array_equal(grad[sel, :20_000].cpu().numpy(), og, equal_nan=True)
    assert (val < 0).any()  # some points are inside the arm


def test_readme_size_link_grids_match_oracle_bitwise():
    """C4 at the reference README's setting (README.md:150-151, tests/test_model_to_sdf.py:182): link caches at
    resolution 0.02 with padding=1.0 -> ~1.3 M voxels (21 MB) per link, 170 MB in total, every workspace point inside
    every link's range.  The fused kernel (large-grid mode: inline exact index fallback) vs the oracle on a slice of
    configurations x points, bit for bit, plus the small-grid mode forced on the same scene."""
    import workloads as Wk
    robot = Wk.build_c4(resolution=0.02, padding=1.0)
    leaves = robot.sdf.sdfs
    assert all(l._packed.shape[0] > 1_000_000 for l in leaves)
    A, P = 6, 20_480
    robot.set_joint_configuration(Wk.c4_joint_configs(A, seed=3))
    pts = Wk.c4_points(P, seed=4)
    val, grad = robot(pts)
    assert val.shape == (A, P) and robot.sdf._query_flags == pv._lib.COMPOSED_INLINE_EXACT
    tfm = robot.object_to_link_frames.get_matrix().cpu().numpy()
    ogrids = [H.oracle_grid_from_cached(l) for l in leaves]
    oval, ograd, oleaf = oracle.composed_query(ogrids, tfm, A, pts.cpu().numpy())
    assert np.array_equal(val.cpu().numpy(), oval, equal_nan=True)
    assert np.array_equal(grad.cpu().numpy(), ograd, equal_nan=True)
    assert len(np.unique(oleaf)) == 8  # every link wins somewhere
    inr = np.mean([l.voxels.get_valid_values(pts).float().mean().item() for l in leaves[:1]])
    assert inr > 0.7  # the regime: most points are inside the padded range of every link
    robot.sdf._query_flags = 0  # the other index mode: same bits
    v0, g0 = robot(pts)
    assert torch.equal(v0, val) and torch.equal(g0.nan_to_num(5.0), grad.nan_to_num(5.0))


@pytest.mark.parametrize("P", [20_480, 20_481, 777])
def test_bucketed_path_returns_the_direct_path_bits_in_caller_order(P):
    """ComposedSDF.bucket_points: Morton-sort the shared point set, run the fused kernel on the sorted points, un-permute
    (pvamd_composed_query_bucketed).  Same bits as the direct path, in the caller's order, for point counts that are and
    are not multiples of the tile / vector widths, on both grid sizes (= both index modes)."""
    import workloads as Wk
    for padding in (0.1, 1.0):
        robot = Wk.build_c4(resolution=0.02, padding=padding)
        A = 9
        robot.set_joint_configuration(Wk.c4_joint_configs(A, seed=5))
        pts = Wk.c4_points(P, seed=6)
        robot.sdf.bucket_points = False
        v_direct, g_direct = robot(pts)
        robot.sdf.bucket_points = True
        v_b, g_b = robot(pts)
        assert v_b.shape == (A, P) and g_b.shape == (A, P, 3)
        assert torch.equal(v_b, v_direct) and torch.equal(g_b.nan_to_num(4.0), g_direct.nan_to_num(4.0))
    robot.sdf.bucket_points = "auto"
    assert robot.sdf._bucketing_pays(200, 1 << 18) and not robot.sdf._bucketing_pays(2, 1 << 18)


@pytest.mark.parametrize("P", [20_480, 20_481, 777, 1])
def test_prepared_points_return_the_direct_path_bits(P):
    """prepare_points once, query_prepared under several joint configurations: order="caller" is __call__'s result (shape and
    bits) without the per-call sort; order="sorted" is the same bits in the handle's order, with neither sort nor un-permute
    pass.  Both grid sizes (= both index modes), point counts on and off the tile width, batched point dims, float64 in."""
    import workloads as Wk
    for padding in (0.1, 1.0):
        robot = Wk.build_c4(resolution=0.02, padding=padding)
        pts = Wk.c4_points(P, seed=6)
        handle = robot.prepare_points(pts)
        assert len(handle) == P and handle.order.dtype == torch.int32
        assert torch.equal(handle.sorted_points, pts[handle.order.long()])
        assert torch.equal(handle.inverse.long()[handle.order.long()], torch.arange(P, device="cuda"))
        for A, seed in ((9, 5), (1, 6), (9, 7)):
            robot.set_joint_configuration(Wk.c4_joint_configs(A, seed=seed) if A > 1 else Wk.c4_joint_configs(2, seed=seed)[1])
            robot.sdf.bucket_points = False
            v_direct, g_direct = robot(pts)
            robot.sdf.bucket_points = "auto"
            v_c, g_c = robot.query_prepared(handle)
            v_s, g_s = robot.query_prepared(handle, order="sorted")
            assert v_c.shape == v_direct.shape and g_c.shape == g_direct.shape
            assert torch.equal(v_c, v_direct) and torch.equal(g_c.nan_to_num(4.0), g_direct.nan_to_num(4.0))
            idx = handle.order.long()
            assert torch.equal(v_s, v_direct[..., idx]) and torch.equal(g_s.nan_to_num(4.0), g_direct[..., idx, :].nan_to_num(4.0))
    if P == 20_480:  # batched point dims and a float64 / host point set come back like __call__'s
        robot.set_joint_configuration(Wk.c4_joint_configs(3, seed=2))
        batched = pts.reshape(10, 2048, 3).double().cpu()
        h2 = robot.prepare_points(batched)
        robot.sdf.bucket_points = False
        v_direct, g_direct = robot(batched.float().cuda())
        robot.sdf.bucket_points = "auto"
        v_c, g_c = robot.query_prepared(h2)
        assert v_c.shape == (3, 10, 2048) and g_c.shape == (3, 10, 2048, 3) and v_c.dtype == torch.float64
        assert torch.equal(v_c.float(), v_direct) and torch.equal(g_c.float().nan_to_num(4.0), g_direct.nan_to_num(4.0))
        v_s, _ = robot.query_prepared(h2, order="sorted")
        assert v_s.shape == (3, 20_480) and torch.equal(v_s.float(), v_direct.reshape(3, -1)[:, h2.order.long()])
        with pytest.raises(ValueError, match="order"):
            robot.query_prepared(h2, order="hilbert")


def test_auto_bucketing_looks_at_what_the_query_can_touch():
    """bucket_points = "auto" with README-size link grids: random points over the workspace are Morton-sorted; a planar slice
    in grid order -- the reference README's own query shape (README.md:177-183), larger -- is not (its cut through each leaf
    grid stays in L2 whatever the order: 0.87 ms direct, 1.25 ms bucketed for 200 x 512 x 512); a 64^3 grid filling the
    volume is.  Either way the results are the direct path's bits."""
    import workloads as Wk
    robot = Wk.build_c4(resolution=0.02, padding=1.0)
    A = 9
    robot.set_joint_configuration(Wk.c4_joint_configs(A, seed=5))
    lo, hi = Wk.ARM_BOX
    ax = [torch.linspace(lo[d], hi[d], 192) for d in range(3)]
    slice_pts = torch.cartesian_prod(ax[0], torch.tensor([0.02]), ax[2]).cuda()   # 36,864 points, one plane
    ax40 = [torch.linspace(lo[d], hi[d], 40) for d in range(3)]
    cube_pts = torch.cartesian_prod(*ax40).cuda()                                  # 64,000 points, the volume
    rand_pts = Wk.c4_points(1 << 16, seed=8)
    comp = robot.sdf
    assert comp.bucket_points == "auto"
    assert not comp._bucketing_pays(A, slice_pts.shape[0], slice_pts)
    assert comp._bucketing_pays(A, cube_pts.shape[0], cube_pts) and comp._bucketing_pays(A, rand_pts.shape[0], rand_pts)
    assert not comp._bucketing_pays(2, rand_pts.shape[0], rand_pts)  # too few configurations to share a sort
    for pts in (slice_pts, cube_pts):
        comp.bucket_points = "auto"
        v_auto, g_auto = robot(pts)
        comp.bucket_points = True
        v_b, g_b = robot(pts)
        assert torch.equal(v_auto, v_b) and torch.equal(g_auto.nan_to_num(4.0), g_b.nan_to_num(4.0))
    comp.bucket_points = "auto"
