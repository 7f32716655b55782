"""RobotSDF: an SDF of an articulated robot, conditioned on (batched) joint configurations
(reference model_to_sdf.py:12-133).  One leaf per mesh visual; forward kinematics gives world_T_link per
configuration; `pvamd_transform_stack` (f32 MFMA) contracts them with the visual offsets into the leaf-major
object->leaf stack that the fused ComposedSDF kernel consumes."""
import ctypes
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
        self._stack, self._stack_obj = None, None  # object_to_link_frames: the (S*A, 4, 4) stack / the Transform3d around it
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
        if joint_config.dim() == 0 or joint_config.shape[-1] != M:
            # (the kernel reads q[a * M + joint]: a short float32 GPU vector, used where it is, would be read out of bounds)
            raise ValueError(f"joint configuration of shape {tuple(joint_config.shape)} does not hold the {M} joint values "
                             f"({', '.join(self.joint_names)}) in its last dimension")
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
                # ONE launch (pvamd_configure_chain): sin / cos + frame walk + offset^-1 o world^-1 on the f32 MFMA.  Joint values
                # that already sit on this GPU as float32 (A, M) are used where they are; anything else is one H2D copy.
                A = 1 if joint_config.dim() == 1 else joint_config.shape[0]
                if joint_config.is_cuda and joint_config.dtype is torch.float32 and joint_config.device == dev and \
                        joint_config.is_contiguous():
                    q = joint_config
                else:
                    q = joint_config.reshape(A, M).to(device=dev, dtype=torch.float32).contiguous()
                stack = self._configure(lib, dev, q, A, M, S, offset_inv)
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
        self._stack, self._stack_obj = stack, None  # the Transform3d is built when object_to_link_frames is read
        if self.sdf is not None:
            self.sdf.set_transforms(stack, batch_dim=self.configuration_batch, known_rigid=True)

    @property
    def object_to_link_frames(self) -> typing.Optional[tf.Transform3d]:
        """model_to_sdf.py:113: the [A*]S object -> link transforms of the current configuration, leaf-major."""
        if self._stack_obj is None and self._stack is not None:
            # configure_and_query_into re-writes its stack in place on the next call: what the caller holds must not change
            # under it (the reference returns a fresh object per configuration)
            own = any(hit[1] is self._stack for hit in self.__dict__.get("_cfg_stack", {}).values())
            self._stack_obj = tf.Transform3d(matrix=self._stack.clone() if own else self._stack)
        return self._stack_obj

    @object_to_link_frames.setter
    def object_to_link_frames(self, value):
        self._stack_obj = value
        self._stack = None if value is None else tf.as_matrix(value)

    def _configure(self, lib, dev, q, A, M, S, offset_inv, stack=None, sincos=None, one_launch_only=False):
        """pvamd_configure_chain on the current stream: q (A, M) float32 on `dev` -> the (S*A, 4, 4) obj->leaf stack.  The
        frame scratch is kept per (device, stream, batch size): it is private to one call on one stream, and two streams that
        configure the same robot concurrently must not share it.  A robot with more SDF-carrying links than the one-launch
        kernel can stage in LDS (PVAMD_E_SHAPE, ~50 links) takes sin / cos + pvamd_chain_fk + pvamd_transform_stack -- the same
        fma chains, any S -- unless the caller needs the single launch (`one_launch_only`: graph capture)."""
        joints, F = self._joint_table_dev(dev)
        stream = _lib.stream_ptr()
        key = (str(dev), stream.value, A)
        hit = self._per_stream("_cfg_scratch", key, lambda: torch.empty((F, 12, A), dtype=torch.float32, device=dev))
        if stack is None:
            stack = torch.empty((S * A, 4, 4), dtype=torch.float32, device=dev)
        rc = lib.pvamd_configure_chain(joints.data_ptr(), F, q.data_ptr(), A, M, offset_inv.data_ptr(), S,
                                       None if sincos is None else sincos.data_ptr(), hit[1].data_ptr(), None,
                                       stack.data_ptr(), stream)
        if rc == _lib.E_SHAPE and not one_launch_only:  # (a shape the three-launch path rejects as well is reported there)
            sin_q, cos_q = torch.sin(q), torch.cos(q)
            if sincos is not None:
                sincos.copy_(torch.stack((sin_q, cos_q), dim=-1))
            link_world = torch.empty((S * A, 4, 4), dtype=torch.float32, device=dev)
            _lib.check(lib.pvamd_chain_fk(joints.data_ptr(), F, q.data_ptr(), sin_q.data_ptr(), cos_q.data_ptr(), A, M,
                                          hit[1].data_ptr(), link_world.data_ptr(), stream), "pvamd_chain_fk")
            _lib.check(lib.pvamd_transform_stack(offset_inv.data_ptr(), link_world.data_ptr(), S, A, stack.data_ptr(), stream),
                       "pvamd_transform_stack")
        elif rc == _lib.E_SHAPE and one_launch_only:
            lds = S * 12 * 64 * 4 + F * ctypes.sizeof(_lib.JointDesc) + 64 * M * 2 * 4
            raise ValueError(f"configure_and_query_into: pvamd_configure_chain refused the shape (frames F={F}, joints M={M}, "
                             f"SDF-carrying links S={S}, configurations A={A}); the one-launch configure kernel needs F, A, S >= 1 and "
                             f"{lds} bytes of LDS (limit 153600: about 50 SDF-carrying links); otherwise call "
                             "set_joint_configuration(q) and query_into(...) instead")
        elif rc != 0:
            _lib.check(rc, "pvamd_configure_chain")
        return stack

    def _per_stream(self, name, key, make):
        """(key, tensor) of a buffer kept per (device, stream, batch size); at most 8 keys are remembered."""
        table = self.__dict__.setdefault(name, {})
        hit = table.get(key)
        if hit is None:
            if len(table) >= 8:
                table.pop(next(iter(table)))
            hit = table[key] = (key, make())
        return hit

    def configure_and_query_into(self, joint_config, points, out_val, out_grad):
        """A planner's inner step without host work between its two launches: joint values (A, M) float32 ALREADY ON THE GPU
        -> pvamd_configure_chain -> pvamd_composed_query into the caller's (A, P) / (A, P, 3) buffers (for large batches on small
        link grids the query is the chunk-grouped pair, ComposedSDF.group_points: one more launch).  No host data, no
        allocation after the first call with this batch size: capturable in a hipGraph (model_to_sdf.py:82-125 as two
        or three kernels).  Needs a kinematics.Chain (on-device FK) and BOUNDING_BOX CachedSDF leaves; afterwards the object is
        configured exactly as by set_joint_configuration(joint_config)."""
        if not hasattr(self.chain, "joint_table"):
            raise ValueError("configure_and_query_into needs the on-device forward kinematics (pytorch_volumetric_amd.kinematics.Chain)")
        M, S = len(self.joint_names), len(self.sdf_to_link_name)
        q = joint_config
        if not (torch.is_tensor(q) and q.is_cuda and q.dtype is torch.float32 and q.dim() == 2 and q.shape[1] == M and q.is_contiguous()):
            raise ValueError(f"configure_and_query_into needs contiguous float32 (A, {M}) joint values on the GPU")
        A = q.shape[0]
        dev = q.device
        lib = _lib.load()
        with _lib.on_device(dev):
            key = (str(dev), _lib.current_raw_stream(dev.index), A)
            # the stack this entry point writes is its own (per stream), re-used call after call
            hit = self._per_stream("_cfg_stack", key, lambda: torch.empty((S * A, 4, 4), dtype=torch.float32, device=dev))
            self._configure(lib, dev, q, A, M, S, self._offset_inv_dev(dev), stack=hit[1], one_launch_only=True)
            self.q, self.configuration_batch = q, (A,)
            if self._stack is not hit[1]:
                self._stack = hit[1]
            self._stack_obj = None  # a Transform3d handed out earlier keeps the configuration it was read under (a copy)
            cur = self.sdf._tf_matrix
            if cur is None or cur.data_ptr() != hit[1].data_ptr() or cur.shape != hit[1].shape or self.sdf.tsf_batch != (A,):
                self.sdf.set_transforms(hit[1], batch_dim=(A,), known_rigid=True)
            self.sdf.invalidate_transforms()  # this stack is re-written in place: drop everything derived from its old contents
            self.sdf.query_into(points, out_val, out_grad)

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

    def prepare_points(self, points_in_object_frame):
        """Sort a re-used query point set once (ComposedSDF.prepare_points); query it under every later joint configuration
        with query_prepared(handle, order="caller" | "sorted")."""
        return self.sdf.prepare_points(points_in_object_frame)

    def query_prepared(self, prepared, order="caller"):
        return self.sdf.query_prepared(prepared, order=order)


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
