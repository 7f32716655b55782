"""Minimal rigid-transform container with the members of pytorch_kinematics.Transform3d that the reference's hot
path touches: get_matrix, inverse, compose, stack, transform_points, transform_normals, len,
slicing, dtype/device.  4x4 matrices, column-vector convention (translation in [:3, 3]).

Every API of this package that takes a transform accepts a raw (..., 4, 4) tensor, one of these, or any object with
a `get_matrix()` method (so a real pytorch_kinematics.Transform3d works unchanged).
"""
import math

import torch


def as_matrix(tsf, dtype=None, device=None):
    """(B,4,4) tensor from a tensor / Transform3d / anything with get_matrix()."""
    if tsf is None:
        return None
    m = tsf.get_matrix() if hasattr(tsf, "get_matrix") else torch.as_tensor(tsf)
    if m.dim() == 2:
        m = m.unsqueeze(0)
    if m.shape[-2:] != (4, 4):
        raise ValueError(f"expected (...,4,4) homogeneous transforms, got {tuple(m.shape)}")
    m = m.reshape(-1, 4, 4)
    if dtype is not None or device is not None:
        m = m.to(dtype=dtype or m.dtype, device=device or m.device)
    return m


def quaternion_to_matrix(q_wxyz):
    """(...,4) unit quaternions, real part first -> (...,3,3)."""
    q = torch.as_tensor(q_wxyz)
    w, x, y, z = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    rows = torch.stack((
        1 - two_s * (y * y + z * z), two_s * (x * y - z * w), two_s * (x * z + y * w),
        two_s * (x * y + z * w), 1 - two_s * (x * x + z * z), two_s * (y * z - x * w),
        two_s * (x * z - y * w), two_s * (y * z + x * w), 1 - two_s * (x * x + y * y)), dim=-1)
    return rows.reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_matrix(axis, angle):
    """Rodrigues: axis (3,) unit, angle (...,) -> (...,3,3)."""
    axis = torch.as_tensor(axis, dtype=angle.dtype, device=angle.device)
    x, y, z = axis.unbind(-1)
    c, s = torch.cos(angle), torch.sin(angle)
    t = 1 - c
    rows = torch.stack((
        t * x * x + c, t * x * y - s * z, t * x * z + s * y,
        t * x * y + s * z, t * y * y + c, t * y * z - s * x,
        t * x * z - s * y, t * y * z + s * x, t * z * z + c), dim=-1)
    return rows.reshape(angle.shape + (3, 3))


def rpy_to_matrix(rpy):
    """URDF fixed-axis roll-pitch-yaw -> 3x3 (R = Rz(yaw) Ry(pitch) Rx(roll))."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return torch.tensor([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                         [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                         [-sp, cp * sr, cp * cr]], dtype=torch.float64)


def rigid_inverse(m):
    """Inverse of (...,4,4) rigid transforms using R^T (no general matrix inverse)."""
    r_t = m[..., :3, :3].transpose(-1, -2)
    out = torch.zeros_like(m)
    out[..., :3, :3] = r_t
    out[..., :3, 3] = -(r_t @ m[..., :3, 3:4]).squeeze(-1)
    out[..., 3, 3] = 1
    return out


def is_rigid(m, tol=1e-5):
    """True when every (...,4,4) matrix is a rigid transform: last row 0 0 0 1 and R R^T = I, det R = +1, to `tol`
    (absolute; rotation matrices built in float32 are orthonormal to ~1e-6, a chain of eight of them to a few 1e-6).
    The tolerance is what the fused kernel's leaf-culling bounds absorb: they carry 1e-4 of relative slack
    (composed.hip build_cull_spheres / tile_leaf_mask), ten times what a matrix accepted here can stretch a distance by."""
    m = torch.as_tensor(m)
    if m.numel() == 0:
        return True
    r = m[..., :3, :3].double()
    last = m[..., 3, :].double()
    eye = torch.eye(3, dtype=torch.float64, device=m.device)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float64, device=m.device)
    ok_last = bool(((last - bottom).abs() <= tol).all())
    ok_rot = bool(((r @ r.transpose(-1, -2) - eye).abs() <= tol).all()) and bool((torch.linalg.det(r) > 0).all())
    return ok_last and ok_rot


class Transform3d:
    def __init__(self, matrix=None, pos=None, rot=None, dtype=torch.float32, device="cpu"):
        """matrix: (B,4,4) or (4,4); or pos (.., 3) and/or rot: (..,3,3) matrix or (..,4) wxyz quaternion."""
        if matrix is not None:
            m = torch.as_tensor(matrix)
            if m.dim() == 2:
                m = m.unsqueeze(0)
            self._matrix = m.reshape(-1, 4, 4)
        else:
            if pos is not None:
                pos = torch.as_tensor(pos, dtype=dtype, device=device).reshape(-1, 3)
            if rot is not None:
                rot = torch.as_tensor(rot, dtype=dtype, device=device)
                rot = quaternion_to_matrix(rot) if rot.shape[-1] == 4 and rot.shape[-2:] != (3, 3) else rot
                rot = rot.reshape(-1, 3, 3)
            n = max(pos.shape[0] if pos is not None else 1, rot.shape[0] if rot is not None else 1)
            m = torch.eye(4, dtype=dtype, device=device).repeat(n, 1, 1)
            if rot is not None:
                m[:, :3, :3] = rot
            if pos is not None:
                m[:, :3, 3] = pos
            self._matrix = m

    # --- the members the hot path uses ---
    def get_matrix(self):
        return self._matrix

    def __len__(self):
        return self._matrix.shape[0]

    def __getitem__(self, item):
        m = self._matrix[item]
        return Transform3d(matrix=m if m.dim() == 3 else m.unsqueeze(0))

    @property
    def dtype(self):
        return self._matrix.dtype

    @property
    def device(self):
        return self._matrix.device

    def to(self, device=None, dtype=None):
        return Transform3d(matrix=self._matrix.to(device=device or self.device, dtype=dtype or self.dtype))

    def inverse(self):
        return Transform3d(matrix=rigid_inverse(self._matrix))

    def compose(self, *others):
        """self followed by others in the pytorch_kinematics sense: a.compose(b) has matrix a @ b."""
        m = self._matrix
        for o in others:
            m = m @ as_matrix(o, dtype=m.dtype, device=m.device)
        return Transform3d(matrix=m)

    def stack(self, *others):
        return Transform3d(matrix=torch.cat([self._matrix] + [as_matrix(o, dtype=self.dtype, device=self.device)
                                                              for o in others], dim=0))

    def transform_points(self, points):
        """(P,3) or (B,P,3) -> (B,P,3); a single transform on (P,3) returns (P,3)."""
        m = self._matrix
        p = points if points.dim() == 3 else points.unsqueeze(0)
        out = p @ m[:, :3, :3].transpose(-1, -2) + m[:, None, :3, 3]
        if points.dim() == 2 and m.shape[0] == 1:
            out = out.squeeze(0)
        return out

    def transform_normals(self, normals):
        m = self._matrix
        n = normals if normals.dim() == 3 else normals.unsqueeze(0)
        out = n @ m[:, :3, :3].transpose(-1, -2)
        if normals.dim() == 2 and m.shape[0] == 1:
            out = out.squeeze(0)
        return out

    def __repr__(self):
        return f"Transform3d(batch={len(self)}, dtype={self.dtype}, device={self.device})"


def Translate(x, y=None, z=None, dtype=torch.float32, device="cpu"):
    pos = torch.as_tensor(x, dtype=dtype, device=device) if y is None else torch.tensor([x, y, z], dtype=dtype,
                                                                                        device=device)
    return Transform3d(pos=pos, dtype=dtype, device=device)


def Rotate(rot, dtype=torch.float32, device="cpu"):
    return Transform3d(rot=rot, dtype=dtype, device=device)


def random_rotations(n, dtype=torch.float32, device="cpu", generator=None):
    """n uniformly distributed rotation matrices from normalised gaussian quaternions."""
    q = torch.randn(n, 4, dtype=torch.float64, generator=generator)
    q = q / q.norm(dim=-1, keepdim=True)
    return quaternion_to_matrix(q).to(dtype=dtype, device=device)
