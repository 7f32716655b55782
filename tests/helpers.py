"""Shared builders for the parity tests: the same synthetic grids / meshes / inputs for the HIP path and the oracle."""
import os

import numpy as np
import torch

import workloads
from oracle import oracle
from pytorch_volumetric_amd import mesh_io

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MESHES = workloads.MESHES

DRILL_BB = np.array([[-0.067981, 0.095006], [-0.041332, 0.081863], [-0.003716, 0.183718]])


mesh_path = workloads.mesh_path


class AnalyticEllipsoidSDF:
    """A cheap closed-form stand-in for a mesh SDF: a sphere SDF in a stretched space (not a true distance, but
    smooth, signed, with a unit gradient) -- only used to fill voxel grids deterministically on the CPU."""

    def __init__(self, center, radii, bb):
        self.center = torch.tensor(center, dtype=torch.float64)
        self.radii = torch.tensor(radii, dtype=torch.float64)
        self.bb = torch.tensor(bb, dtype=torch.float64)

    def __call__(self, pts):
        center, radii = self.center.to(pts.device), self.radii.to(pts.device)
        p = (pts.double() - center) / radii
        n = torch.linalg.norm(p, dim=-1)
        val = (n - 1.0) * radii.min()
        g = p / radii
        g = g / (torch.linalg.norm(g, dim=-1, keepdim=True) + 1e-12)
        return val.to(pts.dtype), g.to(pts.dtype)

    def surface_bounding_box(self, padding=0., padding_ratio=0.):
        bb = self.bb.clone()
        ext = bb[:, 1] - bb[:, 0]
        bb[:, 0] -= padding + padding_ratio * ext
        bb[:, 1] += padding + padding_ratio * ext
        return bb


def drill_like_gt():
    c = DRILL_BB.mean(axis=1)
    r = (DRILL_BB[:, 1] - DRILL_BB[:, 0]) / 2
    return AnalyticEllipsoidSDF(c, r, DRILL_BB)


def padded_range(bb, padding, as_numpy=True):
    r = np.array(bb, dtype=np.float64)
    r[:, 0] -= padding
    r[:, 1] += padding
    return r if as_numpy else [(float(a), float(b)) for a, b in r]


def oracle_grid_from_cached(cached, oob_mode=None):
    """Build the oracle's grid (reference layout: separate val / grad arrays) from a CachedSDF instance."""
    packed = cached._packed.cpu().numpy()
    view = cached._view
    val = packed[:, 0].reshape(view.shape)
    grad = packed[:, 1:4]
    if view.index_f64:
        rmin, rmax = view.dmin.numpy().astype(np.float64), view.dmax.numpy().astype(np.float64)
    else:
        rmin, rmax = view.fmin.numpy().astype(np.float32), view.fmax.numpy().astype(np.float32)
    bb = cached.bb.cpu().numpy()
    if oob_mode is None:
        oob_mode = 1 if cached.out_of_bounds_strategy.value == 1 else 0
    return oracle.Grid(val, grad, rmin, rmax, bb, oob_mode=oob_mode, index_f64=view.index_f64, rule=view.rule)


def oracle_mesh_from_factory(obj):
    m = obj._mesh
    return oracle.Mesh(m.triangle_soup().astype(np.float32), obj._face_normals.astype(np.float32),
                       obj.bounding_box(padding=1.0)[:, 1])


random_rigid = workloads.random_rigid
uniform_points = workloads.uniform_points


def composed_disagreements_explained(leaves, tf, A, pts, val_a, val_b, ulps=8, atol=1e-6, report=None):
    """Two evaluations of the same ComposedSDF that round the obj->leaf transform differently (torch matmul vs the
    kernel's fma chain) must agree to `atol` EXCEPT where a last-place difference in a leaf-frame coordinate changes a
    discrete decision: the coordinate sits within `ulps` units of a half-voxel plane -- the nearest-voxel index flips --
    or of an edge of the cached range -- the in-range / bounding-box branch flips.  One unit = 2^-24 x (|m0 px| + |m1 py|
    + |m2 pz| + |m3|), the bound of ONE float32 rounding of that coordinate; each evaluation makes four (three products
    folded by two fmas and an add), so two evaluations differ by at most 8 units: that is the default, and what the
    error analysis gives (round 2 allowed 16).  `report`, a dict, receives the largest number of units any disagreement
    actually needed.  Returns (number of disagreeing (config, point) pairs, how many of them no plane or edge explains)."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    tf = np.asarray(tf, dtype=np.float64).reshape(len(leaves), A, 4, 4)
    va, vb = np.asarray(val_a, dtype=np.float64).reshape(A, -1), np.asarray(val_b, dtype=np.float64).reshape(A, -1)
    bad = ~(np.isclose(va, vb, rtol=0, atol=atol) | (np.isnan(va) & np.isnan(vb)))
    need = np.full(bad.shape, np.inf)  # units to the nearest plane / edge, over leaves
    for s, leaf in enumerate(leaves):
        v = leaf._view
        mn, mx, res = v.dmin.numpy(), v.dmax.numpy(), v.dres.numpy()
        for a in range(A):
            if not bad[a].any():
                continue
            R, t = tf[s, a, :3, :3], tf[s, a, :3, 3]
            x = pts @ R.T + t
            unit = 2.0 ** -24 * (np.abs(pts) @ np.abs(R).T + np.abs(t))  # per coordinate
            cell = (x - mn) / res
            to_plane = np.abs(cell - np.floor(cell) - 0.5) * res
            to_edge = np.minimum(np.abs(x - mn), np.abs(x - mx))
            near = ((x >= mn - 64 * unit) & (x <= mx + 64 * unit)).all(axis=1)  # the leaf's range is in play at all
            d = np.where(near[:, None], np.minimum(to_plane, to_edge) / unit, np.inf).min(axis=1)
            need[a] = np.minimum(need[a], d)
    explained = need <= ulps
    if report is not None:
        report["max_units_needed"] = float(need[bad].max()) if bad.any() else 0.0
    return int(bad.sum()), int((bad & ~explained).sum())
