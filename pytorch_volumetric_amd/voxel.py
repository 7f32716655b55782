"""Voxel-grid definition helpers and the value-range view used by CachedSDF.

Mirrors the reference's grid layout (pytorch_volumetric/voxel.py:10-25): ranges are snapped to a whole number of
voxels, coordinates are fp32 aranges, and points enumerate the grid in C order (x slowest, z fastest).
"""
import numpy as np
import torch

from pytorch_volumetric_amd import _lib


def get_divisible_range_by_resolution(resolution, range_per_dim):
    """Snap every (low, high) to (low, low + n*resolution) with n = round(span/resolution) (voxel.py:10-17).

    Python's round (half to even) on a python/numpy float, exactly as the reference; the element types of the
    input are preserved, because they decide the dtype of the index arithmetic downstream (see RangeView)."""
    snapped = []
    for bounds in range_per_dim:
        low, high = bounds[0], bounds[1]
        n_cells = round((high - low) / resolution)
        snapped.append((low, low + n_cells * resolution))
    return snapped


def get_coordinates_and_points_in_grid(resolution, range_per_dim, dtype=torch.float, device='cpu', get_points=True):
    """Per-axis voxel-centre coordinates and their cartesian product (voxel.py:20-25)."""
    coords = []
    for bounds in range_per_dim:
        low, high = bounds[0], bounds[1]
        # the 0.9*resolution slack makes the upper end inclusive without admitting one extra cell
        coords.append(torch.arange(low, high + 0.9 * resolution, resolution, dtype=dtype, device=device))
    pts = torch.cartesian_prod(*coords) if get_points else None
    return coords, pts


class RangeView:
    """The numbers a value-range view of a dense grid needs: min, max, resolution per dim, in the dtype torch would
    give them.

    The reference wraps its cache in multidim_indexing's TorchMultidimView (sdf.py:521), which builds
    `torch.tensor([low...])`, `torch.tensor([high...])` and `(max - min) / (torch.tensor(shape) - 1)` from the
    range it is handed.  torch infers float64 for numpy float64 scalars (ranges coming from
    `obj.bounding_box(...)`) and float32 for python floats, and every later `(points - min) / resolution` promotes
    accordingly.  This class reproduces those three tensors with the same torch calls so that the device kernels
    can do the index arithmetic in the same dtype (pvamd_grid_t.index_f64).
    """

    def __init__(self, range_per_dim, shape):
        lows = [b[0] for b in range_per_dim]
        highs = [b[1] for b in range_per_dim]
        vmin = torch.tensor(lows)
        vmax = torch.tensor(highs)
        if not vmin.dtype.is_floating_point:
            vmin, vmax = vmin.to(torch.get_default_dtype()), vmax.to(torch.get_default_dtype())
        if vmin.dtype not in (torch.float32, torch.float64):
            vmin, vmax = vmin.float(), vmax.float()
        self.shape = tuple(int(s) for s in shape)
        self.index_f64 = vmin.dtype == torch.float64
        cells = torch.tensor(self.shape) - 1
        self.min, self.max = vmin, vmax
        self.resolution = (vmax - vmin) / cells
        # both triples are always available to the kernels
        self.dmin, self.dmax = vmin.double(), vmax.double()
        self.dres = self.resolution.double() if not self.index_f64 else self.resolution
        self.fmin, self.fmax = vmin.float(), vmax.float()
        self.fres = self.resolution.float() if self.index_f64 else self.resolution

    def fill(self, desc: "_lib.GridDesc"):
        for d in range(3):
            desc.dmin[d], desc.dmax[d], desc.dres[d] = self.dmin[d].item(), self.dmax[d].item(), self.dres[d].item()
            desc.fmin[d], desc.fmax[d], desc.fres[d] = self.fmin[d].item(), self.fmax[d].item(), self.fres[d].item()
            desc.shape[d] = self.shape[d]
        desc.index_f64 = 1 if self.index_f64 else 0


def bounds_contain_another_bounds(outer_bounds, inner_bounds):
    """Whether outer_bounds (d x 2) contains inner_bounds (voxel.py:134-136)."""
    outer_bounds, inner_bounds = np.asarray(outer_bounds), np.asarray(inner_bounds)
    return bool(np.all(outer_bounds[:, 0] <= inner_bounds[:, 0]) and np.all(outer_bounds[:, 1] >= inner_bounds[:, 1]))


# ---------------------------------------------------------------------------------------------------------------------
# Voxel containers (reference voxel.py:28-171).  Not on the SDF-query hot path (SURVEY.md section 2 marks them out of
# scope; section 8(f) ranks them last of the "next" rows): kept THIN -- plain torch on whatever device the caller
# uses, d = 2 or 3 -- over the same value-range indexing rule the query kernels implement.
# ---------------------------------------------------------------------------------------------------------------------
import abc
import copy
import math


class ValueRangeView:
    """Dense d-dimensional storage addressed by real-valued coordinates: index = round_half_even((p - min)/res) with
    res = (max - min)/(shape - 1), valid iff min <= p <= max (the TorchMultidimView members the reference uses)."""

    def __init__(self, source, value_ranges, invalid_value=0):
        self.device, self.dtype = source.device, source.dtype
        self.shape = tuple(source.shape)
        self.raw_data = source.reshape(-1)
        self._min = torch.tensor([b[0] for b in value_ranges], device=self.device)
        self._max = torch.tensor([b[1] for b in value_ranges], device=self.device)
        cells = (torch.tensor(self.shape, device=self.device) - 1).clamp_min(1)
        self._resolution = (self._max - self._min) / cells
        self.invalid_value = invalid_value

    def ensure_index_key(self, key):
        return torch.round((key - self._min) / self._resolution).to(torch.long)

    def ensure_value_key(self, index):
        return index.to(self._resolution.dtype) * self._resolution + self._min

    def ravel_multi_index(self, key, shape=None):
        shape = shape or self.shape
        flat = torch.zeros(key.shape[:-1], dtype=torch.long, device=key.device)
        for d in range(len(shape)):
            flat = flat * shape[d] + key[..., d]
        return flat

    def get_valid_values(self, key):
        return ((self._min <= key) & (key <= self._max)).all(dim=-1)

    def __getitem__(self, pts):
        valid = self.get_valid_values(pts)
        flat = self.ravel_multi_index(self.ensure_index_key(pts).clamp_min(0).minimum(
            torch.tensor(self.shape, device=self.device) - 1))
        out = self.raw_data[flat]
        fill = self.invalid_value(pts[~valid]) if callable(self.invalid_value) else self.invalid_value
        out = out.clone()
        out[~valid] = fill
        return out

    def __setitem__(self, pts, value):
        valid = self.get_valid_values(pts)
        flat = self.ravel_multi_index(self.ensure_index_key(pts[valid]))
        self.raw_data[flat] = value[valid] if torch.is_tensor(value) and value.dim() > 0 else value


class Voxels(abc.ABC):
    @abc.abstractmethod
    def get_known_pos_and_values(self):
        """positions (N x d) and values (N) of known voxels"""

    @abc.abstractmethod
    def __getitem__(self, pts):
        """values (N) at positions (N x d)"""

    @abc.abstractmethod
    def __setitem__(self, pts, value):
        """set values (N) at positions (N x d)"""


class VoxelGrid(Voxels):
    def __init__(self, resolution, range_per_dim, dtype=torch.float, device='cpu'):
        self.resolution = resolution
        self.invalid_val = 0
        self.dtype = dtype
        self.device = device
        self._create_voxels(resolution, range_per_dim)

    def _create_voxels(self, resolution, range_per_dim):
        self.range_per_dim = get_divisible_range_by_resolution(resolution, range_per_dim)
        self.coords, self.pts = get_coordinates_and_points_in_grid(resolution, self.range_per_dim, device=self.device)
        self._data = torch.zeros([len(c) for c in self.coords], dtype=self.dtype, device=self.device)
        self.voxels = ValueRangeView(self._data, self.range_per_dim, invalid_value=self.invalid_val)
        self.range_per_dim = np.array(self.range_per_dim)

    def get_known_pos_and_values(self):
        known = self.voxels.raw_data != self.invalid_val
        flat = known.nonzero().squeeze(-1)
        idx = torch.stack(torch.unravel_index(flat, self.voxels.shape), dim=-1)
        return self.voxels.ensure_value_key(idx), self.voxels.raw_data[flat]

    def resize_to_fit(self):
        known_pos, known_val = self.get_known_pos_and_values()
        if known_pos.numel() == 0:
            return
        lo, hi = known_pos.min(dim=0).values, known_pos.max(dim=0).values
        range_per_dim = copy.deepcopy(self.range_per_dim)
        for d in range(len(lo)):
            range_per_dim[d] = (lo[d].item() - self.resolution, hi[d].item() + self.resolution)
        self._create_voxels(self.resolution, range_per_dim)
        self.__setitem__(known_pos, known_val)

    def get_voxel_values(self):
        return self._data

    def get_voxel_center_points(self):
        return self.pts

    def __getitem__(self, pts):
        return self.voxels[pts]

    def __setitem__(self, pts, value):
        self.voxels[pts] = value


class ExpandingVoxelGrid(VoxelGrid):
    def __setitem__(self, pts, value):
        if pts.numel() > 0:
            lo, hi = pts.min(dim=0).values, pts.max(dim=0).values
            range_per_dim = copy.deepcopy(self.range_per_dim)
            for d in range(len(lo)):
                over = (hi[d] - self.range_per_dim[d][1]).item()
                under = (self.range_per_dim[d][0] - lo[d]).item()
                if over > 0:
                    range_per_dim[d][1] += math.ceil(over / self.resolution) * self.resolution
                if under > 0:
                    range_per_dim[d][0] -= math.ceil(under / self.resolution) * self.resolution
            if not np.allclose(range_per_dim, self.range_per_dim):
                known_pos, known_values = self.get_known_pos_and_values()
                self._create_voxels(self.resolution, range_per_dim)
                super().__setitem__(known_pos, known_values)
        return super().__setitem__(pts, value)


class VoxelSet(Voxels):
    def __init__(self, positions, values):
        self.positions = positions
        self.values = values

    def __getitem__(self, pts):
        raise RuntimeError("Cannot get arbitrary points on a voxel set")

    def __setitem__(self, pts, value):
        self.positions = torch.cat((self.positions, pts.view(-1, self.positions.shape[-1])), dim=0)
        self.values = torch.cat((self.values, value))

    def get_known_pos_and_values(self):
        return self.positions, self.values


def voxel_down_sample(points, resolution, range_per_dim=None, ignore_flat_dim=False):
    """Replace a point cloud by the centres of the occupied cells of a voxel grid (voxel.py:139-171)."""
    if points.shape[0] == 0:
        return points
    data_bounds = np.stack((points.min(dim=0)[0].cpu().numpy() - resolution * 2,
                            points.max(dim=0)[0].cpu().numpy() + resolution * 2)).T
    if range_per_dim is None or bounds_contain_another_bounds(range_per_dim, data_bounds):
        range_per_dim = data_bounds
    flat_z = ignore_flat_dim and range_per_dim[-1][0] == range_per_dim[-1][1]
    flat_z_val = range_per_dim[-1][0]
    if flat_z:
        range_per_dim = range_per_dim[:-1]
        points = points[..., :-1]
    device = points.device
    voxel = VoxelGrid(resolution, range_per_dim, device=device, dtype=torch.bool)
    voxel[points] = 1
    pts, _ = voxel.get_known_pos_and_values()
    if flat_z:
        pts = torch.cat((pts, torch.ones((pts.shape[0], 1), device=device) * flat_z_val), dim=-1)
    return pts
