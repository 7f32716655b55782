"""RobotSDF: an SDF of an articulated robot, conditioned on (batched) joint configurations
(reference model_to_sdf.py:12-133).  One leaf per mesh visual; forward kinematics gives world_T_link per
configuration; `pvamd_transform_stack` (f32 MFMA) contracts them with the visual offsets into the leaf-major
object->leaf stack that the fused ComposedSDF kernel consumes."""
import logging
import math
import typing

import numpy as np
import torch

from pytorch_volumetric_amd import _lib
from pytorch_volumetric_amd import sdf
from pytorch_volumetric_amd import transforms as tf

logger = logging.getLogger(__file__)


class RobotSDF(sdf.ObjectFrameSDF):
    """SDF for a robot model described by a kinematic chain (pytorch_volumetric_amd.kinematics.Chain or a
    pytorch_kinematics.Chain); the joint configuration must be set before querying."""

    def __init__(self, chain, default_joint_config=None, path_prefix='',
                 link_sdf_cls: typing.Callable[[sdf.ObjectFactory], sdf.ObjectFrameSDF] = sdf.MeshSDF):
        """
        :param chain: robot description; links with non-mesh visuals are ignored (with a warning)
        :param default_joint_config: joint values used until set_joint_configuration is called; None = zeros
        :param path_prefix: prefix for the relative mesh paths found in the robot description
        :param link_sdf_cls: factory of each link's SDF from its ObjectFactory (MeshSDF, or cache_link_sdf_factory())
        """
        self.chain = chain
        self.dtype = self.chain.dtype
        self.device = self.chain.device
        self.q = None
        self.object_to_link_frames: typing.Optional[tf.Transform3d] = None
        self.joint_names = self.chain.get_joint_parameter_names()
        self.frame_names = self.chain.get_frame_names(exclude_fixed=False)
        self.sdf: typing.Optional[sdf.ComposedSDF] = None
        self.sdf_to_link_name = []
        self.configuration_batch = None

        sdfs, offsets = self._collect_mesh_links(path_prefix, link_sdf_cls)
        self.offset_transforms = tf.Transform3d(matrix=torch.cat([tf.as_matrix(o) for o in offsets], dim=0)).to(
            device=self.device, dtype=self.dtype)
        self.sdf = sdf.ComposedSDF(sdfs, self.object_to_link_frames)
        self.set_joint_configuration(default_joint_config)

    def _collect_mesh_links(self, path_prefix, link_sdf_cls):
        """One leaf SDF (+ its visual offset) per mesh visual of the chain, in frame order (model_to_sdf.py:41-56)."""
        sdfs, offsets = [], []
        for link in (self.chain.find_frame(name).link for name in self.frame_names):
            for visual in link.visuals:
                if visual.geom_type != "mesh":
                    logger.warning(f"Cannot handle non-mesh link visual type {visual} for {link.name}")
                    continue
                mesh_file, mesh_scale = visual.geom_param[0], visual.geom_param[1]
                sdfs.append(link_sdf_cls(sdf.MeshObjectFactory(mesh_file, scale=mesh_scale, path_prefix=path_prefix)))
                offsets.append(visual.offset)
                self.sdf_to_link_name.append(link.name)
        if not sdfs:
            raise RuntimeError("robot description has no mesh visuals to build an SDF from")
        return sdfs, offsets

    def surface_bounding_box(self, **kwargs):
        return self.sdf.surface_bounding_box(**kwargs)

    def link_bounding_boxes(self):
        """[A x] S x 8 x 3 corner points of each link's bounding box in the robot frame under the current
        configuration (model_to_sdf.py:65-80)."""
        tfs = tf.Transform3d(matrix=tf.rigid_inverse(tf.as_matrix(self.sdf.obj_frame_to_link_frame)))
        bbs = []
        for i in range(len(self.sdf.sdfs)):
            bb = aabb_to_ordered_end_points(np.asarray(self.sdf.sdfs[i].surface_bounding_box(padding=0)))
            bb = torch.tensor(bb, device=tfs.device, dtype=tfs.dtype)
            bbs.append(tfs[self.sdf.ith_transform_slice(i)].transform_points(bb))
        return torch.stack(bbs).squeeze()

    def set_joint_configuration(self, joint_config=None):
        """
        :param joint_config: [A x] M joint values; A may be any number of batch dimensions (model_to_sdf.py:82-115)
        """
        M = len(self.joint_names)
        if joint_config is None:
            joint_config = torch.zeros(M, device=self.device, dtype=self.dtype)
        joint_config = torch.as_tensor(joint_config)
        if len(joint_config.shape) > 1:
            self.configuration_batch = tuple(joint_config.shape[:-1])
            # a chain of fixed joints only has M == 0: reshape(-1, 0) cannot infer the batch
            joint_config = joint_config.reshape(math.prod(self.configuration_batch), M)
        else:
            self.configuration_batch = None
        self.q = joint_config
        S = len(self.sdf_to_link_name)
        lib = _lib.load()
        dev = _lib.require_gpu()
        offset_inv = self._offset_inv_dev(dev)
        with _lib.on_device(dev):
            if hasattr(self.chain, "joint_table"):
                # on-device FK (pvamd_chain_fk): no per-frame host-driven ops, nothing returns to the host
                q = joint_config.reshape(1 if joint_config.dim() == 1 else joint_config.shape[0], M).to(
                    device=dev, dtype=torch.float32).contiguous()
                A = q.shape[0]
                joints, F = self._joint_table_dev(dev)
                sin_q, cos_q = torch.sin(q), torch.cos(q)
                scratch = torch.empty((F, 12, A), dtype=torch.float32, device=dev)
                link_world_d = torch.empty((S * A, 4, 4), dtype=torch.float32, device=dev)
                _lib.check(lib.pvamd_chain_fk(_lib.ptr(joints), F, _lib.ptr(q), _lib.ptr(sin_q), _lib.ptr(cos_q), A, M,
                                              _lib.ptr(scratch), _lib.ptr(link_world_d), _lib.stream_ptr()),
                           "pvamd_chain_fk")
            else:
                # a foreign chain object (e.g. pytorch_kinematics.Chain): use its own forward kinematics
                fk = self.chain.forward_kinematics(joint_config, end_only=False)
                link_world = torch.cat([tf.as_matrix(fk[name]) for name in self.sdf_to_link_name])  # leaf-major
                A = link_world.shape[0] // S
                link_world_d = link_world.to(device=dev, dtype=torch.float32).contiguous()
            # object_to_link[s*A+a] = offset[s]^-1 @ world_T_link[s,a]^-1 on the matrix cores
            stack = torch.empty_like(link_world_d)
            _lib.check(lib.pvamd_transform_stack(_lib.ptr(offset_inv), _lib.ptr(link_world_d), S, A,
                                                 _lib.ptr(stack), _lib.stream_ptr()), "pvamd_transform_stack")
        self.object_to_link_frames = tf.Transform3d(matrix=stack)
        if self.sdf is not None:
            self.sdf.set_transforms(self.object_to_link_frames, batch_dim=self.configuration_batch, known_rigid=True)

    def _offset_inv_dev(self, dev):
        if getattr(self, "_offset_inv_cache", None) is None or self._offset_inv_cache.device != dev:
            self._offset_inv_cache = tf.rigid_inverse(self.offset_transforms.get_matrix()).to(
                device=dev, dtype=torch.float32).contiguous()
        return self._offset_inv_cache

    def _joint_table_dev(self, dev):
        if getattr(self, "_joint_cache", None) is None or self._joint_cache[0].device != dev:
            raw = self.chain.joint_table(self.sdf_to_link_name)
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
            import ctypes
            self._joint_cache = (t, len(raw) // ctypes.sizeof(_lib.JointDesc))
        return self._joint_cache

    def __call__(self, points_in_object_frame):
        """
        :param points_in_object_frame: [B x] N x 3 points in the robot frame
        :return: [A x] [B x] N values and [A x] [B x] N x 3 gradients (A = configuration batch dims)
        """
        return self.sdf(points_in_object_frame)

    def query_into(self, points, out_val, out_grad):
        """Allocation-free form of __call__ (see ComposedSDF.query_into); outputs are (A,P) and (A,P,3)."""
        self.sdf.query_into(points, out_val, out_grad)


def cache_link_sdf_factory(resolution=0.01, padding=0.1, **kwargs):
    """model_to_sdf.py:128-133"""

    def create_sdf(obj_factory: sdf.ObjectFactory):
        gt_sdf = sdf.MeshSDF(obj_factory)
        return sdf.CachedSDF(obj_factory.name, resolution, obj_factory.bounding_box(padding=padding), gt_sdf, **kwargs)

    return create_sdf


_CORNER_ORDER = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1))
_SEQUENTIAL_ORDER = ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 0, 0),
                     (1, 0, 1), (1, 1, 1), (1, 1, 0), (1, 1, 1), (0, 1, 1), (0, 1, 0), (0, 1, 1), (0, 0, 1))


def aabb_to_ordered_end_points(aabb, arrange_in_sequential_order=False):
    """8 corners (or the 16-vertex wireframe polyline) of a (3,2) [[min,max],...] box, in the reference's order
    (model_to_sdf.py:136-171)."""
    order = _SEQUENTIAL_ORDER if arrange_in_sequential_order else _CORNER_ORDER
    arr = [[aabb[d, pick[d]] for d in range(3)] for pick in order]
    if torch.is_tensor(aabb):
        return torch.tensor(arr, device=aabb.device, dtype=aabb.dtype)
    return np.array(arr)
