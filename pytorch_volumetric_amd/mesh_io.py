"""Triangle-mesh container and file loading (host side, numpy float64).

Stands in for the legacy open3d TriangleMesh the reference loads at sdf.py:103-113: vertices are float64, faces are
triangles (polygons are fan-triangulated), and the frame operations (scale, rotate about the origin, translate) are
applied in float64 before anything is rounded to the float32 the query kernels use.
"""
import os

import numpy as np


class TriMesh:
    def __init__(self, vertices, faces):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, dtype=np.int64).reshape(-1, 3)
        if self.faces.size and (self.faces.min() < 0 or self.faces.max() >= len(self.vertices)):
            raise ValueError("face index out of range")

    def copy(self):
        return TriMesh(self.vertices.copy(), self.faces.copy())

    def scaled(self, scale):
        """uniform (scalar) or per-axis (3-vector) scale about the origin"""
        return TriMesh(self.vertices * np.asarray(scale, dtype=np.float64), self.faces)

    def rotated(self, rot3x3):
        return TriMesh(self.vertices @ np.asarray(rot3x3, dtype=np.float64).T, self.faces)

    def translated(self, offset):
        return TriMesh(self.vertices + np.asarray(offset, dtype=np.float64).reshape(1, 3), self.faces)

    def triangle_soup(self):
        """[F,3,3] float64 corner positions."""
        return self.vertices[self.faces]

    def triangle_normals(self):
        """[F,3] float64 unit normals, (v1-v0) x (v2-v0) normalised; zero-area faces get a zero normal."""
        t = self.triangle_soup()
        n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        length = np.linalg.norm(n, axis=1, keepdims=True)
        return np.divide(n, length, out=np.zeros_like(n), where=length > 0)

    def triangle_areas(self):
        t = self.triangle_soup()
        return 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)

    def aabb(self):
        return self.vertices.min(axis=0), self.vertices.max(axis=0)

    def center(self):
        return self.vertices.mean(axis=0)


def morton_order(points, bits=10):
    """Permutation that sorts 3-D points along a Z-order curve (2^bits cells per axis over their bounding box)."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    if len(p) == 0:
        return np.zeros((0,), dtype=np.int64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    cell = np.clip(((p - lo) / np.maximum(hi - lo, 1e-30) * (2 ** bits - 1)).astype(np.int64), 0, 2 ** bits - 1)
    key = np.zeros(len(p), dtype=np.int64)
    for b in range(bits):
        for d in range(3):
            key |= ((cell[:, d] >> b) & 1) << (3 * b + d)
    return np.argsort(key, kind="stable")


def _parse_obj(text):
    verts, faces = [], []
    for line in text.splitlines():
        if not line or line[0] not in "vf":
            continue
        parts = line.split()
        if parts[0] == "v":
            verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
        elif parts[0] == "f":
            idx = []
            for tok in parts[1:]:
                i = int(tok.split("/")[0])
                idx.append(i - 1 if i > 0 else len(verts) + i)  # negative = relative to the current end
            for k in range(1, len(idx) - 1):
                faces.append((idx[0], idx[k], idx[k + 1]))
    return np.array(verts, dtype=np.float64).reshape(-1, 3), np.array(faces, dtype=np.int64).reshape(-1, 3)


def _parse_stl(data):
    if data[:5].lower() == b"solid" and b"facet" in data[:1024]:
        verts = []
        for line in data.decode("ascii", "ignore").splitlines():
            parts = line.split()
            if len(parts) == 4 and parts[0] == "vertex":
                verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
        v = np.array(verts, dtype=np.float64).reshape(-1, 3)
    else:
        n = int(np.frombuffer(data, dtype="<u4", count=1, offset=80)[0])
        rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n,
                            offset=84)
        v = rec["v"].reshape(-1, 3).astype(np.float64)
    return v, np.arange(len(v), dtype=np.int64).reshape(-1, 3)


def load_mesh(path):
    """.obj (text), .stl (ascii/binary) or .npz with arrays `vertices` [V,3] and `faces` [F,3]."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        with np.load(path) as z:
            return TriMesh(z["vertices"], z["faces"])
    if ext == ".stl":
        with open(path, "rb") as f:
            return TriMesh(*_parse_stl(f.read()))
    with open(path, "r") as f:
        return TriMesh(*_parse_obj(f.read()))


def save_obj(path, mesh):
    with open(path, "w") as f:
        for v in mesh.vertices:
            f.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        for t in mesh.faces:
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")


# ---- procedural meshes (closed, outward-oriented) used by tests and the synthetic benchmark configs ----
def box_mesh(half_extents=(1.0, 1.0, 1.0), center=(0.0, 0.0, 0.0)):
    h = np.asarray(half_extents, dtype=np.float64)
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) * h
    faces = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
                      [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int64)
    return TriMesh(corners + np.asarray(center, dtype=np.float64), faces)


def uv_sphere_mesh(radius=1.0, n_lon=32, n_lat=16, scale=(1.0, 1.0, 1.0), center=(0.0, 0.0, 0.0)):
    """Lat-long sphere: 2*n_lon*(n_lat-1) triangles.  n_lon=250, n_lat=200 -> 99,500 (BASELINE config C5)."""
    verts = [(0.0, 0.0, radius)]
    for i in range(1, n_lat):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            verts.append((radius * np.sin(th) * np.cos(ph), radius * np.sin(th) * np.sin(ph), radius * np.cos(th)))
    verts.append((0.0, 0.0, -radius))
    south = len(verts) - 1
    faces = []
    ring = lambda i, j: 1 + (i - 1) * n_lon + (j % n_lon)
    for j in range(n_lon):
        faces.append((0, ring(1, j), ring(1, j + 1)))
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            faces.append((a, c, d))
            faces.append((a, d, b))
    for j in range(n_lon):
        faces.append((south, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)))
    v = np.array(verts, dtype=np.float64) * np.asarray(scale, dtype=np.float64) + np.asarray(center, dtype=np.float64)
    return TriMesh(v, np.array(faces, dtype=np.int64))
