"""Mutable voxel containers and point-cloud down-sampling: the API of the reference's voxel.py:28-171.

SURVEY.md section 8(f) rank 4: the scatter twin of the query kernels' gather.  A 3-D float32 / bool grid that lives on
the GPU, read or written with float32 points, goes through the HIP kernels `pvamd_voxel_gather_* / pvamd_voxel_scatter_*`
(csrc/voxelgrid.hip: the same index arithmetic as the SDF kernels, "last point in input order wins" for points sharing
a voxel).  Everything else -- other dimensionalities (the reference down-samples 2-D clouds too), other dtypes, CPU
tensors -- is plain torch over the same value-range indexing rule (index = round_half_even((p - min) / res), valid
iff min <= p <= max); that generic path is container bookkeeping, not an SDF query, and nothing on the query path uses it.
"""
import abc
import ctypes

import numpy as np
import torch

from pytorch_volumetric_amd import _lib
from pytorch_volumetric_amd.voxel import (RangeView, bounds_contain_another_bounds, get_coordinates_and_points_in_grid,
                                          get_divisible_range_by_resolution)


class ValueRangeView:
    """Dense d-dimensional storage addressed by real-valued coordinates."""

    def __init__(self, source, value_ranges, invalid_value=0):
        self.device, self.dtype = source.device, source.dtype
        self.shape = tuple(source.shape)
        self.raw_data = source.reshape(-1)
        self._min = torch.tensor([b[0] for b in value_ranges], device=self.device)
        self._max = torch.tensor([b[1] for b in value_ranges], device=self.device)
        self._extent = torch.tensor(self.shape, device=self.device)
        self.invalid_value = invalid_value
        self._ranges = [(b[0], b[1]) for b in value_ranges]
        self._desc = None
        from pytorch_volumetric_amd import voxel
        self._rule = voxel.INDEX_RULE  # the view's unpinned choices (voxel.INDEX_RULE), as of construction
        cells = (self._extent - 1).clamp_min(1)
        self._resolution = (self._max - self._min) / cells
        if (self._rule & _lib.RULE_RES_F64) and self._min.dtype == torch.float32:
            # the same statement as voxel.RangeView (what the kernels' descriptor holds): evaluated in float64, then cast
            self._resolution = ((self._max.double() - self._min.double()) / cells).float()

    # ---- HIP path: 3-D float32 / bool storage on the GPU, float32 points ----
    def _device_path(self, pts):
        return (self.raw_data.is_cuda and len(self.shape) == 3 and min(self.shape) >= 2
                and self.dtype in (torch.float32, torch.bool) and torch.is_tensor(pts) and pts.is_cuda
                and pts.dtype == torch.float32 and pts.shape[-1] == 3 and self.raw_data.is_contiguous())

    def _grid_desc(self):
        if self._desc is None:
            desc = _lib.GridDesc()
            RangeView(self._ranges, self.shape, rule=self._rule).fill(desc)  # same dtype inference as the torch tensors above
            desc.oob_mode = _lib.OOB_BOUNDING_BOX
            _lib.check(_lib.load().pvamd_grid_finalize(ctypes.byref(desc)), "pvamd_grid_finalize")
            self._desc = desc
        return self._desc

    def _device_get(self, pts):
        lib = _lib.load()
        flat = pts.reshape(-1, 3).contiguous()
        P = flat.shape[0]
        as_bytes = self.dtype == torch.bool
        store = self.raw_data.view(torch.uint8) if as_bytes else self.raw_data
        out = torch.empty((P,), dtype=store.dtype, device=flat.device)
        invalid = self.invalid_value
        fn = lib.pvamd_voxel_gather_u8 if as_bytes else lib.pvamd_voxel_gather_f32
        with _lib.on_device(flat.device):
            _lib.check(fn(ctypes.byref(self._grid_desc()), _lib.ptr(store), _lib.ptr(flat), P,
                          int(bool(invalid)) if as_bytes else float(invalid), _lib.ptr(out), _lib.stream_ptr()),
                       "pvamd_voxel_gather")
        out = out.view(torch.bool) if as_bytes else out
        return out.reshape(pts.shape[:-1])

    def _device_set(self, pts, value):
        lib = _lib.load()
        flat = pts.reshape(-1, 3).contiguous()
        P = flat.shape[0]
        if P == 0:
            return
        as_bytes = self.dtype == torch.bool
        store = self.raw_data.view(torch.uint8) if as_bytes else self.raw_data
        per_point = torch.is_tensor(value) and value.dim() > 0
        values, owner, scalar = None, None, 0
        if per_point:
            values = value.reshape(-1).to(device=flat.device, dtype=self.dtype).contiguous()
            values = values.view(torch.uint8) if as_bytes else values
            owner = torch.empty((self.raw_data.numel(),), dtype=torch.int32, device=flat.device)
        else:
            scalar = value.item() if torch.is_tensor(value) else value
        fn = lib.pvamd_voxel_scatter_u8 if as_bytes else lib.pvamd_voxel_scatter_f32
        with _lib.on_device(flat.device):
            _lib.check(fn(ctypes.byref(self._grid_desc()), _lib.ptr(store), _lib.ptr(flat), _lib.ptr(values),
                          int(bool(scalar)) if as_bytes else float(scalar), P, _lib.ptr(owner), _lib.stream_ptr()),
                       "pvamd_voxel_scatter")

    def _rounded(self, key):
        q = (key - self._min) / self._resolution
        if self._rule & _lib.RULE_ROUND_HALF_AWAY:
            # roundf as the kernels state it: q - trunc(q) is exact, so no sum that could round up (|q| = 0.49999997 -> 0)
            whole = torch.trunc(q)
            return whole + torch.where((q - whole).abs() >= 0.5, torch.sign(q), torch.zeros_like(q))
        if self._rule & _lib.RULE_ROUND_FLOOR_HALF:
            return torch.floor(q + 0.5)
        return torch.round(q)

    def ensure_index_key(self, key):
        return self._rounded(key).to(torch.long)

    def ensure_value_key(self, index):
        return index.to(self._resolution.dtype) * self._resolution + self._min

    def ravel_multi_index(self, key, shape=None):
        flat = key[..., 0].clone()
        for d, n in enumerate((shape or self.shape)[1:], start=1):
            flat = flat * n + key[..., d]
        return flat

    def get_valid_values(self, key):
        if self._rule & _lib.RULE_VALID_ON_INDEX:
            k = self._rounded(key)
            return ((k >= 0) & (k <= self._extent - 1)).all(dim=-1)
        return ((key >= self._min) & (key <= self._max)).all(dim=-1)

    def _flat(self, pts):
        return self.ravel_multi_index(self.ensure_index_key(pts).clamp_min(0).minimum(self._extent - 1))

    def __getitem__(self, pts):
        if self._device_path(pts) and not callable(self.invalid_value):
            return self._device_get(pts)
        ok = self.get_valid_values(pts)
        out = self.raw_data[self._flat(pts)].clone()
        out[~ok] = self.invalid_value(pts[~ok]) if callable(self.invalid_value) else self.invalid_value
        return out

    def __setitem__(self, pts, value):
        if self._device_path(pts):
            return self._device_set(pts, value)
        ok = self.get_valid_values(pts)
        per_point = torch.is_tensor(value) and value.dim() > 0
        self.raw_data[self._flat(pts[ok])] = value[ok] if per_point else value


class Voxels(abc.ABC):
    """positions <-> values container protocol"""

    @abc.abstractmethod
    def get_known_pos_and_values(self):
        ...

    @abc.abstractmethod
    def __getitem__(self, pts):
        ...

    @abc.abstractmethod
    def __setitem__(self, pts, value):
        ...


class VoxelGrid(Voxels):
    """Dense grid over a fixed range; 0 means "unknown"."""

    def __init__(self, resolution, range_per_dim, dtype=torch.float, device='cpu'):
        self.resolution, self.dtype, self.device = resolution, dtype, device
        self.invalid_val = 0
        self._create_voxels(resolution, range_per_dim)

    def _create_voxels(self, resolution, range_per_dim):
        snapped = get_divisible_range_by_resolution(resolution, range_per_dim)
        # the centre points (a cartesian product the size of the grid) are formed only when asked for
        self.coords, _ = get_coordinates_and_points_in_grid(resolution, snapped, device=self.device, get_points=False)
        self._pts = None
        self._data = torch.zeros(tuple(len(c) for c in self.coords), dtype=self.dtype, device=self.device)
        self.voxels = ValueRangeView(self._data, snapped, invalid_value=self.invalid_val)
        self.range_per_dim = np.array(snapped)

    def get_known_pos_and_values(self):
        idx = (self._data != self.invalid_val).nonzero()  # (K, d) multi-indices
        return self.voxels.ensure_value_key(idx), self._data[tuple(idx.unbind(-1))]

    def _rebuild(self, new_range):
        pos, val = self.get_known_pos_and_values()
        self._create_voxels(self.resolution, new_range)
        if pos.numel():
            self.voxels[pos] = val

    def resize_to_fit(self):
        """shrink / move the range to the known voxels plus one cell of margin"""
        pos, _ = self.get_known_pos_and_values()
        if pos.numel():
            lo, hi = pos.amin(dim=0).cpu().numpy(), pos.amax(dim=0).cpu().numpy()
            self._rebuild(np.stack((lo - self.resolution, hi + self.resolution), axis=1))

    def get_voxel_values(self):
        return self._data

    @property
    def pts(self):
        if self._pts is None:
            self._pts = torch.cartesian_prod(*self.coords)
        return self._pts

    def get_voxel_center_points(self):
        return self.pts

    def __getitem__(self, pts):
        return self.voxels[pts]

    def __setitem__(self, pts, value):
        self.voxels[pts] = value


class ExpandingVoxelGrid(VoxelGrid):
    """VoxelGrid whose range grows, in whole cells, to contain whatever is written to it."""

    def __setitem__(self, pts, value):
        if pts.numel():
            lo, hi = _point_bounds(pts.reshape(-1, pts.shape[-1]))
            cells_below = np.ceil(np.maximum(self.range_per_dim[:, 0] - lo, 0) / self.resolution)
            cells_above = np.ceil(np.maximum(hi - self.range_per_dim[:, 1], 0) / self.resolution)
            if cells_below.any() or cells_above.any():
                self._rebuild(np.stack((self.range_per_dim[:, 0] - cells_below * self.resolution,
                                        self.range_per_dim[:, 1] + cells_above * self.resolution), axis=1))
        super().__setitem__(pts, value)


class VoxelSet(Voxels):
    """Sparse list of (position, value) pairs; append-only."""

    def __init__(self, positions, values):
        self.positions, self.values = positions, values

    def __getitem__(self, pts):
        raise RuntimeError("Cannot get arbitrary points on a voxel set")

    def __setitem__(self, pts, value):
        self.positions = torch.cat((self.positions, pts.reshape(-1, self.positions.shape[-1])))
        self.values = torch.cat((self.values, value))

    def get_known_pos_and_values(self):
        return self.positions, self.values


def _point_bounds(points):
    """(lo, hi) numpy rows of an N x D cloud; one fused kernel for float32 3-D clouds on the GPU"""
    if points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 3:
        flat = points.contiguous()
        box = torch.empty((2, 3), dtype=torch.float32, device=points.device)
        with _lib.on_device(points.device):
            _lib.check(_lib.load().pvamd_points_aabb(_lib.ptr(flat), flat.shape[0], _lib.ptr(box), _lib.stream_ptr()),
                       "pvamd_points_aabb")
        box = box.cpu().numpy()
        if np.isfinite(box).all():  # a cloud with non-finite coordinates takes the plain reductions below
            return box[0], box[1]
    return points.amin(dim=0).cpu().numpy(), points.amax(dim=0).cpu().numpy()


def voxel_down_sample(points, resolution, range_per_dim=None, ignore_flat_dim=False):
    """Centres of the occupied cells of a `resolution` grid laid over an N x D point cloud.

    range_per_dim: optional (D,2) range; used as given unless it contains the data's own (2-cell padded) bounds, in
    which case the data bounds are used.  ignore_flat_dim: a last dimension with min == max is carried through."""
    if len(points) == 0:
        return points
    pad = 2 * resolution
    lo, hi = _point_bounds(points)
    data_bounds = np.stack((lo - pad, hi + pad), axis=1)
    if range_per_dim is None or bounds_contain_another_bounds(range_per_dim, data_bounds):
        range_per_dim = data_bounds
    flat_value = None
    if ignore_flat_dim and range_per_dim[-1][0] == range_per_dim[-1][1]:
        flat_value = range_per_dim[-1][0]
        range_per_dim, points = range_per_dim[:-1], points[..., :-1]
    occupancy = VoxelGrid(resolution, range_per_dim, device=points.device, dtype=torch.bool)
    occupancy[points] = 1
    centres, _ = occupancy.get_known_pos_and_values()
    if flat_value is not None:
        centres = torch.cat((centres, torch.full((len(centres), 1), float(flat_value), device=points.device)), dim=-1)
    return centres
