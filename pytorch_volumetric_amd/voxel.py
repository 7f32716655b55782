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


# Which of the UNPINNED choices of multidim_indexing's value-range view the kernels restate (include/pvamd.h
# PVAMD_RULE_*, _lib.RULE_*): 0 = index by round-half-to-even, a point is valid when min <= p <= max.  Nothing in the
# reference pins these (the package is an un-vendored dependency, call sites sdf.py:521,537-540); should the real
# package turn out to test validity on the rounded index, round halves away from zero / as floor(q + 0.5), or evaluate
# a float32 range's resolution in float64, set this (before building caches) instead of touching a kernel:
#   import pytorch_volumetric_amd as pv; pv.voxel.INDEX_RULE = pv.RULE_VALID_ON_INDEX | pv.RULE_ROUND_HALF_AWAY
# tools/rule_exposure.py measures how many results each alternative changes (profiles/r03_rule_exposure.txt).
INDEX_RULE = 0


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

    def __init__(self, range_per_dim, shape, rule=None):
        self.rule = INDEX_RULE if rule is None else int(rule)
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
        if (self.rule & _lib.RULE_RES_F64) and not self.index_f64:
            self.resolution = ((vmax.double() - vmin.double()) / cells).float()
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
        desc.rule = self.rule


def bounds_contain_another_bounds(outer_bounds, inner_bounds):
    """Whether outer_bounds (d x 2) contains inner_bounds (voxel.py:134-136)."""
    outer_bounds, inner_bounds = np.asarray(outer_bounds), np.asarray(inner_bounds)
    return bool(np.all(outer_bounds[:, 0] <= inner_bounds[:, 0]) and np.all(outer_bounds[:, 1] >= inner_bounds[:, 1]))


_CONTAINERS = ("Voxels", "VoxelGrid", "ExpandingVoxelGrid", "VoxelSet", "voxel_down_sample", "ValueRangeView")


def __getattr__(name):
    """The reference keeps its voxel containers in voxel.py (voxel.py:28-171); here they live in voxel_containers.py (which
    imports this module), and `from pytorch_volumetric_amd.voxel import VoxelGrid` still finds them."""
    if name in _CONTAINERS:
        from pytorch_volumetric_amd import voxel_containers
        return getattr(voxel_containers, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
