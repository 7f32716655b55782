"""numpy-facing loader for oracle/libpvamd_oracle.so (the C restatement of the reference's hot path).

TEST INFRASTRUCTURE ONLY: the checker for the HIP kernels and the timed CPU baseline in bench.py.  Parity status of
the oracle itself: third-party arithmetic UNPINNED (the reference cannot be imported here); see the header of
pvamd_oracle.c and DESIGN.md section "Oracle".
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpvamd_oracle.so")


class OracleGrid(ctypes.Structure):
    _fields_ = [
        ("val", ctypes.c_void_p), ("grad", ctypes.c_void_p),
        ("dmin", ctypes.c_double * 3), ("dmax", ctypes.c_double * 3), ("dres", ctypes.c_double * 3),
        ("fmin", ctypes.c_float * 3), ("fmax", ctypes.c_float * 3), ("fres", ctypes.c_float * 3),
        ("bb_min", ctypes.c_float * 3), ("bb_max", ctypes.c_float * 3),
        ("shape", ctypes.c_int32 * 3), ("index_f64", ctypes.c_int32), ("oob_mode", ctypes.c_int32),
        ("rule", ctypes.c_int32),
        ("dbb_min", ctypes.c_double * 3), ("dbb_max", ctypes.c_double * 3),
    ]


class OracleMesh(ctypes.Structure):
    _fields_ = [("tri", ctypes.c_void_p), ("normal", ctypes.c_void_p), ("F", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("ray_dir", ctypes.c_double * 3)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return load().oracle_num_threads()


def set_num_threads(n):
    load().oracle_set_num_threads(int(n))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


class Grid:
    """A cached grid in the REFERENCE's layout: val [nx,ny,nz] + grad [n,3] (sdf.py:504-505), plus the view numbers.

    range_min / range_max: length-3 sequences; their numpy dtype (float32 / float64) selects the index dtype the way
    torch promotion does in the reference."""

    def __init__(self, val, grad, range_min, range_max, bb, oob_mode=1, index_f64=None, rule=0):
        self.val = _f32(val)
        self.shape = tuple(self.val.shape)
        self.grad = _f32(grad).reshape(-1, 3)
        rmin, rmax = np.asarray(range_min), np.asarray(range_max)
        if index_f64 is None:
            index_f64 = rmin.dtype == np.float64
        self.index_f64 = bool(index_f64)
        g = OracleGrid()
        g.val, g.grad = self.val.ctypes.data, self.grad.ctypes.data
        cells = np.array(self.shape, dtype=np.int64) - 1
        if self.index_f64:
            dmin, dmax = rmin.astype(np.float64), rmax.astype(np.float64)
            dres = (dmax - dmin) / cells
            fmin, fmax, fres = dmin.astype(np.float32), dmax.astype(np.float32), dres.astype(np.float32)
        else:
            fmin, fmax = rmin.astype(np.float32), rmax.astype(np.float32)
            if rule & 8:  # the resolution of a float32 range evaluated in float64, then rounded
                fres = ((fmax.astype(np.float64) - fmin.astype(np.float64)) / cells).astype(np.float32)
            else:
                fres = ((fmax - fmin) / cells.astype(np.float32)).astype(np.float32)
            dmin, dmax, dres = fmin.astype(np.float64), fmax.astype(np.float64), fres.astype(np.float64)
        bb64 = np.asarray(bb, dtype=np.float64).reshape(3, 2)  # float64 queries see the un-rounded box (sdf.py:556-557)
        bb = bb64.astype(np.float32)
        for d in range(3):
            g.dbb_min[d], g.dbb_max[d] = bb64[d, 0], bb64[d, 1]
            g.dmin[d], g.dmax[d], g.dres[d] = dmin[d], dmax[d], dres[d]
            g.fmin[d], g.fmax[d], g.fres[d] = fmin[d], fmax[d], fres[d]
            g.bb_min[d], g.bb_max[d] = bb[d, 0], bb[d, 1]
            g.shape[d] = self.shape[d]
        g.index_f64 = int(self.index_f64)
        g.oob_mode = int(oob_mode)
        g.rule = int(rule)
        self.rule = int(rule)
        self.c = g


def voxel_index(grid, pts):
    pts = _f32(pts).reshape(-1, 3)
    P = len(pts)
    key = np.empty((P, 3), np.int64)
    flat = np.empty((P,), np.int64)
    valid = np.empty((P,), np.uint8)
    load().oracle_voxel_index(ctypes.byref(grid.c), _p(pts), ctypes.c_int64(P), _p(key), _p(flat), _p(valid))
    return key, flat, valid.astype(bool)


def voxel_gather(grid, storage, pts, invalid_value=0):
    """VoxelGrid.__getitem__ (voxel.py:97-98 over the value-range view): storage[flat] where the point is within the
    range, invalid_value elsewhere.  storage: dense C-order array of the grid's shape."""
    _, flat, valid = voxel_index(grid, pts)
    store = np.asarray(storage).reshape(-1)
    out = np.full((len(flat),), invalid_value, dtype=store.dtype)
    out[valid] = store[flat[valid]]
    return out


def voxel_scatter(grid, storage, pts, value):
    """VoxelGrid.__setitem__ (voxel.py:100-103): in-range points write, the rest are ignored; points sharing a voxel
    are applied in input order (numpy index assignment: the last one wins, as in a sequential loop).  In place."""
    _, flat, valid = voxel_index(grid, pts)
    store = np.asarray(storage).reshape(-1)
    if np.ndim(value) > 0:
        store[flat[valid]] = np.asarray(value).reshape(-1)[valid]
    else:
        store[flat[valid]] = value
    return storage


def cached_query(grid, pts):
    pts = _f32(pts).reshape(-1, 3)
    P = len(pts)
    val = np.empty((P,), np.float32)
    grad = np.empty((P, 3), np.float32)
    oob = np.empty((P,), np.uint8)
    load().oracle_cached_query(ctypes.byref(grid.c), _p(pts), ctypes.c_int64(P), _p(val), _p(grad), _p(oob))
    return val, grad, oob.astype(bool)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def voxel_index_f64(grid, pts):
    """float64 query points: index arithmetic and range test in float64 (torch promotion, sdf.py:537-540)."""
    pts = _f64(pts).reshape(-1, 3)
    P = len(pts)
    key, flat, valid = np.empty((P, 3), np.int64), np.empty((P,), np.int64), np.empty((P,), np.uint8)
    load().oracle_voxel_index_f64(ctypes.byref(grid.c), _p(pts), ctypes.c_int64(P), _p(key), _p(flat), _p(valid))
    return key, flat, valid.astype(bool)


def cached_query_f64(grid, pts):
    """CachedSDF.__call__ for float64 query points: float64 outputs (sdf.py:545-547)."""
    pts = _f64(pts).reshape(-1, 3)
    P = len(pts)
    val, grad, oob = np.empty((P,), np.float64), np.empty((P, 3), np.float64), np.empty((P,), np.uint8)
    load().oracle_cached_query_f64(ctypes.byref(grid.c), _p(pts), ctypes.c_int64(P), _p(val), _p(grad), _p(oob))
    return val, grad, oob.astype(bool)


def cached_outside_f64(grid, pts, level=0.0):
    pts = _f64(pts).reshape(-1, 3)
    out = np.empty((len(pts),), np.uint8)
    load().oracle_cached_outside_f64(ctypes.byref(grid.c), _p(pts), ctypes.c_int64(len(pts)), ctypes.c_double(level),
                                     _p(out))
    return out.astype(bool)


def cached_outside(grid, pts, level=0.0):
    pts = _f32(pts).reshape(-1, 3)
    out = np.empty((len(pts),), np.uint8)
    load().oracle_cached_outside(ctypes.byref(grid.c), _p(pts), ctypes.c_int64(len(pts)), ctypes.c_float(level), _p(out))
    return out.astype(bool)


def sample_surface(tri, cdf, n, seed):
    """The counter-based draw of sample_mesh_points: (points [n,3] float64, triangle index [n], subset keys [n])."""
    tri = _f32(tri).reshape(-1, 9)
    cdf = _f64(cdf)
    pts, face, key = np.empty((n, 3), np.float64), np.empty((n,), np.int32), np.empty((n,), np.int64)
    load().oracle_sample_surface(_p(tri), _p(cdf), ctypes.c_int32(len(tri)), ctypes.c_int64(n), ctypes.c_uint64(seed),
                                 _p(pts), _p(face), _p(key))
    return pts, face, key


def transform_points(tf, pts):
    """x[a][p] = tf[a] p, the k-ordered fma chain of the composed kernels (sdf.py:399).  tf: [A,4,4]."""
    tf = _f32(tf).reshape(-1, 16)
    pts = _f32(pts).reshape(-1, 3)
    out = np.empty((len(tf), len(pts), 3), np.float32)
    load().oracle_transform_points(_p(tf), ctypes.c_int32(len(tf)), _p(pts), ctypes.c_int64(len(pts)), _p(out))
    return out


def compose_merge(tf, leaf_val, leaf_grad, s, best_val, best_grad, best_leaf=None):
    """Fold leaf s into the running first minimum in place (sdf.py:409,421); s == 0 initialises."""
    tf = _f32(tf).reshape(-1, 16)
    A, P = leaf_val.shape
    lv, lg = _f32(leaf_val), _f32(leaf_grad)
    load().oracle_compose_merge(_p(tf), ctypes.c_int32(A), ctypes.c_int64(P), _p(lv), _p(lg), ctypes.c_int32(s),
                                ctypes.c_int32(1 if s == 0 else 0), _p(best_val), _p(best_grad), _p(best_leaf))


def composed_query(grids, tf, A, pts):
    """grids: list of S Grid; tf: [S*A,4,4] obj->leaf leaf-major; returns val [A,P], grad [A,P,3], leaf [A,P]."""
    S = len(grids)
    arr = (OracleGrid * S)(*[g.c for g in grids])
    tf = _f32(tf).reshape(S * A, 16)
    pts = _f32(pts).reshape(-1, 3)
    P = len(pts)
    val = np.empty((A, P), np.float32)
    grad = np.empty((A, P, 3), np.float32)
    leaf = np.empty((A, P), np.int32)
    load().oracle_composed_query(arr, ctypes.c_int32(S), _p(tf), ctypes.c_int32(A), _p(pts), ctypes.c_int64(P), _p(val),
                                 _p(grad), _p(leaf))
    return val, grad, leaf


def composed_query_f64(grids, tf, A, pts):
    """float64 points and a float64 [S*A,4,4] transform stack: val [A,P], grad [A,P,3] in float64, leaf [A,P]."""
    S = len(grids)
    arr = (OracleGrid * S)(*[g.c for g in grids])
    tf = np.ascontiguousarray(tf, dtype=np.float64).reshape(S * A, 16)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    P = len(pts)
    val = np.empty((A, P), np.float64)
    grad = np.empty((A, P, 3), np.float64)
    leaf = np.empty((A, P), np.int32)
    load().oracle_composed_query_f64(arr, ctypes.c_int32(S), _p(tf), ctypes.c_int32(A), _p(pts), ctypes.c_int64(P), _p(val),
                                     _p(grad), _p(leaf))
    return val, grad, leaf


class Mesh:
    def __init__(self, tri, normal, ray_dir):
        self.tri = _f32(tri).reshape(-1, 3, 3)
        self.normal = _f32(normal).reshape(-1, 3)
        m = OracleMesh()
        m.tri, m.normal, m.F = self.tri.ctypes.data, self.normal.ctypes.data, len(self.tri)
        for d in range(3):
            m.ray_dir[d] = float(ray_dir[d])
        self.c = m


def mesh_query(mesh, pts, seed=0, index_base=0):
    pts = _f32(pts).reshape(-1, 3)
    P = len(pts)
    closest = np.empty((P, 3), np.float32)
    dist = np.empty((P,), np.float32)
    grad = np.empty((P, 3), np.float32)
    face = np.empty((P,), np.int32)
    normal = np.empty((P, 3), np.float32)
    load().oracle_mesh_query(ctypes.byref(mesh.c), _p(pts), ctypes.c_int64(P), ctypes.c_uint64(seed),
                             ctypes.c_int64(index_base), _p(closest), _p(dist), _p(grad), _p(face), _p(normal))
    return closest, dist, grad, face, normal


def jitter_dir(mesh, seed, index):
    out = np.empty((3,), np.float32)
    load().oracle_jitter_dir(ctypes.byref(mesh.c), ctypes.c_uint64(seed), ctypes.c_int64(index), _p(out))
    return out


def chamfer_mesh(mesh, W, pts, scale=1000.0):
    W = _f32(W).reshape(-1, 16)
    pts = _f32(pts).reshape(-1, 3)
    out = np.empty((len(W),), np.float64)
    load().oracle_chamfer_mesh(ctypes.byref(mesh.c), _p(W), ctypes.c_int32(len(W)), _p(pts), ctypes.c_int64(len(pts)),
                               ctypes.c_float(scale), _p(out))
    return out


def chamfer_grid(grid, W, pts, scale=1000.0):
    W = _f32(W).reshape(-1, 16)
    pts = _f32(pts).reshape(-1, 3)
    out = np.empty((len(W),), np.float64)
    load().oracle_chamfer_grid(ctypes.byref(grid.c), _p(W), ctypes.c_int32(len(W)), _p(pts), ctypes.c_int64(len(pts)),
                               ctypes.c_float(scale), _p(out))
    return out


def transform_stack(offset_inv, link_world, S, A):
    offset_inv = _f32(offset_inv).reshape(S, 16)
    link_world = _f32(link_world).reshape(S * A, 16)
    out = np.empty((S * A, 4, 4), np.float32)
    load().oracle_transform_stack(_p(offset_inv), _p(link_world), ctypes.c_int32(S), ctypes.c_int32(A), _p(out))
    return out


class OracleJoint(ctypes.Structure):
    _fields_ = [("parent", ctypes.c_int32), ("jtype", ctypes.c_int32), ("jcol", ctypes.c_int32),
                ("leaf_slot", ctypes.c_int32), ("axis", ctypes.c_float * 3), ("reserved", ctypes.c_float),
                ("origin", ctypes.c_float * 12)]


def chain_fk(joint_table, q, n_leaves):
    """joint_table: bytes of F pvamd_joint_t records (Chain.joint_table()); q: [A,M] float32.
    Returns (world [F,A,3,4], link_world [S*A,4,4])."""
    F = len(joint_table) // ctypes.sizeof(OracleJoint)
    joints = (OracleJoint * F).from_buffer_copy(joint_table)
    q = _f32(q).reshape(-1, max(1, np.asarray(q).shape[-1]))
    A, M = q.shape
    sq, cq = np.sin(q).astype(np.float32), np.cos(q).astype(np.float32)
    world = np.zeros((F, A, 12), np.float32)
    link_world = np.zeros((max(1, n_leaves) * A, 4, 4), np.float32)
    load().oracle_chain_fk(joints, ctypes.c_int32(F), _p(q), _p(sq), _p(cq), ctypes.c_int32(A), ctypes.c_int32(M),
                           _p(world), _p(link_world))
    return world.reshape(F, A, 3, 4), link_world, sq, cq
