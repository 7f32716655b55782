"""A small URDF kinematic chain with batched forward kinematics -- the slice of pytorch_kinematics.Chain that
RobotSDF consumes (reference model_to_sdf.py:28-35,41-56,99): dtype/device, joint parameter names, frame names,
find_frame(name).link.{name, visuals[*].{geom_type, geom_param, offset}}, forward_kinematics(q, end_only=False).

pytorch_kinematics is neither vendored nor installable here; RobotSDF duck-types on those members, so a real
pk.Chain can be passed instead of this class.
"""
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np
import torch

from pytorch_volumetric_amd import _lib
from pytorch_volumetric_amd import transforms as tf


class Visual:
    def __init__(self, offset: tf.Transform3d, geom_type: Optional[str], geom_param):
        self.offset = offset
        self.geom_type = geom_type
        self.geom_param = geom_param

    def __repr__(self):
        return f"Visual(geom_type={self.geom_type}, geom_param={self.geom_param})"


class Link:
    def __init__(self, name, visuals=()):
        self.name = name
        self.visuals = list(visuals)


class Joint:
    def __init__(self, name, joint_type="fixed", axis=(0.0, 0.0, 1.0), origin=None, limits=None):
        self.name = name
        self.joint_type = joint_type  # fixed | revolute | continuous | prismatic
        self.axis = np.asarray(axis, dtype=np.float64)
        n = np.linalg.norm(self.axis)
        if n > 0:
            self.axis = self.axis / n
        self.origin = origin if origin is not None else np.eye(4)  # parent link frame -> joint frame, float64
        self.limits = limits


class Frame:
    def __init__(self, link: Link, joint: Joint, parent: Optional[int]):
        self.name = link.name
        self.link = link
        self.joint = joint
        self.parent = parent


def _origin_matrix(elem):
    m = np.eye(4)
    if elem is None:
        return m
    xyz = [float(v) for v in elem.get("xyz", "0 0 0").split()]
    rpy = [float(v) for v in elem.get("rpy", "0 0 0").split()]
    m[:3, :3] = tf.rpy_to_matrix(rpy).numpy()
    m[:3, 3] = xyz
    return m


class Chain:
    """Kinematic tree stored in topological order (parents before children)."""

    def __init__(self, frames: List[Frame], dtype=torch.float32, device="cpu"):
        self._frames = frames
        self.dtype = dtype
        self.device = torch.device(device)
        self._by_name: Dict[str, Frame] = {f.name: f for f in frames}

    def to(self, dtype=None, device=None):
        return Chain(self._frames, dtype=dtype or self.dtype, device=device or self.device)

    def get_joint_parameter_names(self, exclude_fixed=True):
        return [f.joint.name for f in self._frames if not (exclude_fixed and f.joint.joint_type == "fixed")]

    def get_frame_names(self, exclude_fixed=True):
        return [f.name for f in self._frames if not (exclude_fixed and f.joint.joint_type == "fixed")]

    def find_frame(self, name):
        return self._by_name.get(name)

    @property
    def n_joints(self):
        return len(self.get_joint_parameter_names())

    def joint_table(self, leaf_links=()):
        """The chain as an array of pvamd_joint_t records (bytes) for the on-device FK kernel; `leaf_links[s]` = name of
        the link whose world matrix goes to slot s of the leaf-major output."""
        import ctypes
        slots = {}
        for s, name in enumerate(leaf_links):
            slots.setdefault(name, []).append(s)
        table = []
        col = 0
        for f in self._frames:
            jt = f.joint.joint_type
            code = {"fixed": 0, "revolute": 1, "continuous": 1, "prismatic": 2}.get(jt)
            if code is None:
                raise ValueError(f"unsupported joint type {jt}")
            rec = _lib.JointDesc()
            rec.parent = -1 if f.parent is None else f.parent
            rec.jtype = code
            rec.jcol = col if code else -1
            col += 1 if code else 0
            rec.leaf_slot = -1
            for k in range(3):
                rec.axis[k] = float(f.joint.axis[k])
            o = np.asarray(f.joint.origin, dtype=np.float64)
            for k in range(12):
                rec.origin[k] = float(o[k // 4, k % 4])
            table.append((rec, slots.get(f.name, [])))
        # a link with several mesh visuals fills several slots: duplicate its record as a fixed child at identity
        out = []
        index_of = {}
        for i, (rec, sl) in enumerate(table):
            index_of[i] = len(out)
            if rec.parent >= 0:
                rec.parent = index_of[rec.parent]
            rec.leaf_slot = sl[0] if sl else -1
            out.append(rec)
            for extra in sl[1:]:
                dup = _lib.JointDesc()
                dup.parent, dup.jtype, dup.jcol, dup.leaf_slot = index_of[i], 0, -1, extra
                for k in range(12):
                    dup.origin[k] = 1.0 if k % 5 == 0 else 0.0
                out.append(dup)
        arr = (_lib.JointDesc * len(out))(*out)
        return bytes(arr)

    def forward_kinematics(self, th, end_only=False):
        """th: (M,) or (A, M) joint values -> {link name: Transform3d of (A,4,4) world_T_link}."""
        th = torch.as_tensor(th, dtype=self.dtype, device=self.device)
        if th.dim() == 1:
            th = th.unsqueeze(0)
        A = th.shape[0]
        eye = torch.eye(4, dtype=self.dtype, device=self.device).expand(A, 4, 4)
        world: List[torch.Tensor] = []
        col = 0
        for f in self._frames:
            parent = eye if f.parent is None else world[f.parent]
            m = parent @ torch.as_tensor(f.joint.origin, dtype=self.dtype, device=self.device)
            jt = f.joint.joint_type
            if jt != "fixed":
                q = th[:, col]
                col += 1
                motion = torch.eye(4, dtype=self.dtype, device=self.device).repeat(A, 1, 1)
                if jt in ("revolute", "continuous"):
                    motion[:, :3, :3] = tf.axis_angle_to_matrix(f.joint.axis, q)
                elif jt == "prismatic":
                    motion[:, :3, 3] = q.unsqueeze(-1) * torch.as_tensor(f.joint.axis, dtype=self.dtype,
                                                                         device=self.device)
                else:
                    raise ValueError(f"unsupported joint type {jt}")
                m = m @ motion
            world.append(m)
        out = {f.name: tf.Transform3d(matrix=w) for f, w in zip(self._frames, world)}
        if end_only:
            return out[self._frames[-1].name]
        return out


def build_chain_from_urdf(urdf_text: str, end_link_name: Optional[str] = None, root_link_name: Optional[str] = None,
                          dtype=torch.float32, device="cpu") -> Chain:
    """Parse a URDF string into a Chain.  With end_link_name only the serial path root -> end link is kept."""
    root = ET.fromstring(urdf_text)
    links: Dict[str, Link] = {}
    for le in root.findall("link"):
        visuals = []
        for ve in le.findall("visual"):
            offset = tf.Transform3d(matrix=torch.tensor(_origin_matrix(ve.find("origin")), dtype=dtype))
            geom = ve.find("geometry")
            gtype, gparam = None, None
            if geom is not None and len(geom):
                g = geom[0]
                gtype = g.tag
                if gtype == "mesh":
                    scale = g.get("scale")
                    gparam = (g.get("filename"), None if scale is None else np.array([float(v) for v in scale.split()]))
                elif gtype == "box":
                    gparam = [float(v) for v in g.get("size").split()]
                elif gtype == "cylinder":
                    gparam = (float(g.get("radius")), float(g.get("length")))
                elif gtype == "sphere":
                    gparam = float(g.get("radius"))
            visuals.append(Visual(offset, gtype, gparam))
        links[le.get("name")] = Link(le.get("name"), visuals)

    children: Dict[str, list] = {}
    child_names = set()
    for je in root.findall("joint"):
        axis_e = je.find("axis")
        # a joint without <axis>: pytorch_kinematics, which the reference drives (model_to_sdf.py:99), falls back to
        # z = (0, 0, 1) (frame.Joint's default), not the URDF specification's (1, 0, 0); drop-in parity follows pk
        axis = [float(v) for v in axis_e.get("xyz").split()] if axis_e is not None else (0.0, 0.0, 1.0)
        joint = Joint(je.get("name"), je.get("type"), axis, _origin_matrix(je.find("origin")))
        parent, child = je.find("parent").get("link"), je.find("child").get("link")
        children.setdefault(parent, []).append((joint, child))
        child_names.add(child)
    if root_link_name is None:
        roots = [n for n in links if n not in child_names]
        if not roots:
            raise ValueError("URDF has no root link")
        root_link_name = roots[0]

    frames: List[Frame] = []

    def walk(link_name, joint, parent_idx):
        frames.append(Frame(links[link_name], joint, parent_idx))
        me = len(frames) - 1
        for j, c in children.get(link_name, []):
            walk(c, j, me)

    walk(root_link_name, Joint(root_link_name + "_root", "fixed"), None)

    if end_link_name is not None:
        idx = next((i for i, f in enumerate(frames) if f.name == end_link_name), None)
        if idx is None:
            raise ValueError(f"link {end_link_name} not in URDF")
        path = []
        while idx is not None:
            path.append(idx)
            idx = frames[idx].parent
        path.reverse()
        remap = {old: new for new, old in enumerate(path)}
        frames = [Frame(frames[i].link, frames[i].joint, remap.get(frames[i].parent)) for i in path]
    return Chain(frames, dtype=dtype, device=device)


def build_serial_chain_from_urdf(urdf_text, end_link_name, root_link_name=None, dtype=torch.float32, device="cpu"):
    return build_chain_from_urdf(urdf_text, end_link_name=end_link_name, root_link_name=root_link_name, dtype=dtype,
                                 device=device)
