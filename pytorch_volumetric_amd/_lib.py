"""ctypes binding of libpvamd.so (C ABI in include/pvamd.h).

The HIP library is the only compute path of this package: there is no CPU or eager-PyTorch fallback.  Loading
fails loudly if the shared object is missing, and every query raises if no MI355X is visible.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PVAMD_LIB") or os.path.join(_HERE, "csrc", "libpvamd.so")  # PVAMD_LIB: A/B builds (tools/)

ABI_VERSION = 12
OOB_LOOKUP_GT_SDF = 0
OOB_BOUNDING_BOX = 1
COMPOSED_INLINE_EXACT = 1
COMPOSED_OUT_PACKED = 32
COMPOSED_NO_GROUPING = 64
COMPOSED_FORCE_FUSED = 128
RULE_VALID_ON_INDEX = 1
RULE_ROUND_HALF_AWAY = 2
RULE_ROUND_FLOOR_HALF = 4
RULE_RES_F64 = 8
TRI_REC = 24
TRI_TILE = 256
TRI_GROUP = 16


def rec_floats(F):
    """PVAMD_REC_FLOATS(F): records are stored in whole tiles."""
    return ((F + TRI_TILE - 1) // TRI_TILE) * TRI_TILE * TRI_REC


def tiles_floats(F):
    """PVAMD_TILES_FLOATS(F): tile spheres followed by group spheres."""
    return ((F + TRI_TILE - 1) // TRI_TILE) * (4 + 4 * (TRI_TILE // TRI_GROUP))


MESH_SCRATCH_GROUPS = 8192
MESH_SMALL_POINTS = 16384  # PVAMD_MESH_SMALL_POINTS


def mesh_scratch_slots(P):
    """PVAMD_MESH_SCRATCH_SLOTS(P)"""
    groups = (P + 63) // 64
    return groups if groups < MESH_SCRATCH_GROUPS else max(groups // 8, MESH_SCRATCH_GROUPS)


def mesh_scratch_bytes(P):
    """PVAMD_MESH_SCRATCH_BYTES(P)"""
    return 64 + mesh_scratch_slots(P) * (64 * 40 + 8 + 64) + 24 * MESH_SMALL_POINTS


_c_float_p = ctypes.POINTER(ctypes.c_float)


class GridDesc(ctypes.Structure):
    """pvamd_grid_t"""
    _fields_ = [
        ("vox", ctypes.c_void_p),
        ("dmin", ctypes.c_double * 3),
        ("dmax", ctypes.c_double * 3),
        ("dres", ctypes.c_double * 3),
        ("fmin", ctypes.c_float * 3),
        ("fmax", ctypes.c_float * 3),
        ("fres", ctypes.c_float * 3),
        ("bb_min", ctypes.c_float * 3),
        ("bb_max", ctypes.c_float * 3),
        ("shape", ctypes.c_int32 * 3),
        ("index_f64", ctypes.c_int32),
        ("oob_mode", ctypes.c_int32),
        ("finalized", ctypes.c_int32),
        ("vlo", ctypes.c_float * 3),
        ("vhi", ctypes.c_float * 3),
        ("inv32", ctypes.c_float * 3),
        ("err32", ctypes.c_float * 3),
        ("rule", ctypes.c_int32),
        ("dbb_min", ctypes.c_double * 3),
        ("dbb_max", ctypes.c_double * 3),
        ("range_n2", ctypes.c_float),
        ("reserved0", ctypes.c_float),
    ]


class MeshDesc(ctypes.Structure):
    """pvamd_mesh_t"""
    _fields_ = [
        ("normal", ctypes.c_void_p),
        ("rec", ctypes.c_void_p),
        ("tiles", ctypes.c_void_p),
        ("rec_of_face", ctypes.c_void_p),
        ("F", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("ray_dir", ctypes.c_double * 3),
        ("pair_counters", ctypes.c_void_p),
    ]


class JointDesc(ctypes.Structure):
    """pvamd_joint_t"""
    _fields_ = [("parent", ctypes.c_int32), ("jtype", ctypes.c_int32), ("jcol", ctypes.c_int32),
                ("leaf_slot", ctypes.c_int32), ("axis", ctypes.c_float * 3), ("reserved", ctypes.c_float),
                ("origin", ctypes.c_float * 12)]


# name -> (restype, argtypes); the test-suite checks every one of these is exported by the .so and declared in
# include/pvamd.h
SIGNATURES = {
    "pvamd_abi_version": (ctypes.c_int, []),
    "pvamd_build_info": (ctypes.c_char_p, []),
    "pvamd_device_count": (ctypes.c_int, []),
    "pvamd_grid_finalize": (ctypes.c_int, [ctypes.POINTER(GridDesc)]),
    "pvamd_pack_grid": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "pvamd_cached_query": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_cached_outside": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_voxel_index": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_cached_query_f64": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_cached_outside_f64": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_voxel_index_f64": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_composed_query": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "pvamd_composed_query_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_composed_query_packed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                                                   ctypes.c_void_p]),
    "pvamd_unpack_records": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_composed_query_bucketed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                     ctypes.c_void_p]),
    "pvamd_cached_query_kernel": (ctypes.c_int, [ctypes.c_int64]),
    "pvamd_group_chunk_points": (ctypes.c_int64, []),
    "pvamd_group_scratch_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "pvamd_group_points": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_composed_query_grouped": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "pvamd_transform_points": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                              ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_compose_merge": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p]),
    "pvamd_voxel_gather_f32": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                              ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_voxel_gather_u8": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_uint8, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_voxel_scatter_f32": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_float, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_voxel_scatter_u8": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_uint8, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_points_aabb": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_morton_keys": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "pvamd_morton_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_mesh_prepare": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_float,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_mesh_query": (ctypes.c_int, [ctypes.POINTER(MeshDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_uint64,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_mesh_query_unordered": (ctypes.c_int, [ctypes.POINTER(MeshDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64,
                                                  ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p]),
    "pvamd_cache_build": (ctypes.c_int, [ctypes.POINTER(MeshDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_sample_surface": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_chamfer_mesh": (ctypes.c_int, [ctypes.POINTER(MeshDesc), ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "pvamd_chamfer_mesh_flat": (ctypes.c_int, [ctypes.POINTER(MeshDesc), ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_chamfer_grid": (ctypes.c_int, [ctypes.POINTER(GridDesc), ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_chain_fk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_pairwise_min_reduce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_configure_chain": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p]),
    "pvamd_transform_stack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_void_p]),
}

E_SHAPE = -2  # PVAMD_E_SHAPE (include/pvamd.h)
_ERRORS = {-1: "a required pointer is NULL", -2: "a size/shape argument is out of range",
           -3: "a pointer is misaligned", -4: "unknown enum value"}

_lib = None


class PvamdError(RuntimeError):
    pass


def load():
    """Load libpvamd.so once; raise if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PvamdError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C pytorch_volumetric_amd/csrc`) from the repository root. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pvamd_abi_version() != ABI_VERSION:
        raise PvamdError(f"libpvamd ABI {lib.pvamd_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    # an A/B build (tools/build_variant.sh: one translation unit compiled with tuning / counter knobs) names itself; the product
    # path and the tests never load one by accident
    if hasattr(lib, "pvamd_variant"):
        lib.pvamd_variant.restype = ctypes.c_char_p
        what = lib.pvamd_variant().decode()
        if os.environ.get("PVAMD_ALLOW_VARIANT") != "1":
            raise PvamdError(f"{LIB_PATH} is an A/B build ({what}); set PVAMD_ALLOW_VARIANT=1 to time it, or unset PVAMD_LIB")
        import sys
        print(f"[pvamd] A/B build loaded: {what} ({LIB_PATH})", file=sys.stderr)
    _lib = lib
    return lib


def check(code, what):
    if code == 0:
        return
    if code < 0:
        raise PvamdError(f"{what}: invalid argument ({_ERRORS.get(code, code)})")
    raise PvamdError(f"{what}: hipError_t {code}")


_gpu_checked = False
_devices = {}


def require_gpu():
    """The compute device of this package.  No GPU -> loud failure (never a silent CPU path)."""
    global _gpu_checked
    if not _gpu_checked:  # availability cannot change within a process; the check costs several us per call otherwise
        if not torch.cuda.is_available():
            raise PvamdError("pytorch_volumetric_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm; "
                             "torch.cuda.is_available() is False and there is no CPU fallback")
        torch.cuda.current_device()  # torch's lazy CUDA/HIP initialisation, once
        _gpu_checked = True
    index = torch._C._cuda_getDevice()
    dev = _devices.get(index)
    if dev is None:
        dev = _devices[index] = torch.device("cuda", index)
    return dev


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw-handle call: torch.cuda.current_stream()
    builds a Stream object and re-validates the device, ~8 us)."""
    if not _gpu_checked:
        require_gpu()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class on_device:
    """`with on_device(dev):` = torch.cuda.device(dev) without its per-entry validation when dev is already current."""
    __slots__ = ("index", "prev")

    def __init__(self, dev):
        self.index = dev.index if dev.index is not None else torch._C._cuda_getDevice()
        self.prev = -1

    def __enter__(self):
        cur = torch._C._cuda_getDevice()
        if cur != self.index:
            self.prev = cur
            torch.cuda.set_device(self.index)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


_fast = False  # False: not looked for yet; None: absent / disabled


def fastcall():
    """The optional host-call module (csrc/fastcall.cpp -> pytorch_volumetric_amd/_pvamd_fast.so; `make -C csrc fast`,
    built by __graft_entry__.build()): the two allocations + the C-ABI call of the drop-in fast paths without the interpreter
    in between (cached(points) 7.4 -> 5.9 us of host time).  Plumbing only -- the same entry points of the same libpvamd.so,
    by address; None when it has not been built or PVAMD_NO_FASTCALL=1, and the ctypes path runs instead."""
    global _fast
    if _fast is False:
        _fast = None
        if os.environ.get("PVAMD_NO_FASTCALL") != "1":
            try:
                from pytorch_volumetric_amd import _pvamd_fast
                _pvamd_fast.set_error_class(PvamdError)
                _fast = _pvamd_fast
            except ImportError:
                pass
    return _fast


def entry_address(name):
    """Address of a libpvamd.so entry point (what _pvamd_fast calls through)."""
    return ctypes.cast(getattr(load(), name), ctypes.c_void_p).value


# Bumped whenever an attribute that the cached call plans of CachedSDF / ComposedSDF depend on is assigned
# (CachedSDF.__setattr__): a plan remembers the epoch it was built in and is rebuilt when it has moved on.
EPOCH = [0]
current_device_index = torch._C._cuda_getDevice
current_raw_stream = torch._C._cuda_getCurrentRawStream


def same_gpu(out_device, dev):
    """Whether results wanted on `out_device` (a str / torch.device as the reference's `device=` arguments take them)
    can stay where the kernel wrote them, on `dev`."""
    d = torch.device(out_device)
    return d.type == "cuda" and (d.index is None or d.index == dev.index)


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def as_query_points(points, device=None, keep_f64=False):
    """[..., 3] any float dtype / device  ->  (contiguous fp32 [P,3] on the GPU, leading shape, dtype, device).

    `device`: the GPU that owns the grid / mesh the points will be looked up in (the kernel dereferences raw pointers
    of that device, so the points and the launch must be there too); None = the current device.
    `keep_f64`: float64 points stay float64 (the grid lookups have float64 entry points; the mesh query computes in
    float32 like the reference, sdf.py:132)."""
    if not torch.is_tensor(points):
        points = torch.as_tensor(points)
    if points.shape[-1] != 3:
        raise ValueError(f"query points must have last dimension 3, got {tuple(points.shape)}")
    dev = require_gpu() if device is None else device
    lead = points.shape[:-1]
    flat = points.detach().reshape(-1, 3).to(device=dev, dtype=torch.float64 if keep_f64 and points.dtype == torch.float64
                                              else torch.float32).contiguous()
    return flat, lead, points.dtype if points.dtype.is_floating_point else torch.float32, points.device


ORDER_RADIX_SORT_FROM = 3 << 18  # PVAMD_ORDER_RADIX_SORT_FROM of the product build; morton_order_scratch_words() sizes for EITHER sort



_group_chunk = None


def group_chunk_points():
    """Points per chunk of the chunk-grouped composed query (a build constant of the library)."""
    global _group_chunk
    if _group_chunk is None:
        _group_chunk = int(load().pvamd_group_chunk_points())
    return _group_chunk


def group_points(flat):
    """pvamd_group_points over contiguous fp32 (P, 3) GPU points: the scratch (uint8 tensor) pvamd_composed_query_grouped reads."""
    lib = load()
    P = flat.shape[0]
    scratch = torch.empty((int(lib.pvamd_group_scratch_bytes(P)),), dtype=torch.uint8, device=flat.device)
    check(lib.pvamd_group_points(ptr(flat), P, ptr(scratch), stream_ptr()), "pvamd_group_points")
    return scratch

def morton_order_scratch_words(P):
    """PVAMD_MORTON_ORDER_SCRATCH_BYTES(P) / 4"""
    # The library picks the sort by a compile-time threshold the binding cannot see in an A/B build (an env override
    # here let Python size the counting sort's scratch while the C side ran the radix sort past it): size for whichever needs
    # more.  Radix: supergroup totals + tile histograms (<= 512 words per 4096-pair tile), 2 key + 2 index arrays.
    radix = 8 + 4 * P + 512 * ((P + 4095) // 4096)
    counting = 8 + (1 << (21 if P >= (1 << 20) else (18 if P >= (1 << 16) else 15))) + P + 2048
    return max(radix, counting)


def morton_order(points, min_points=2048, want_inverse=False, want_sorted=False):
    """int32 permutation that walks fp32 [P,3] device points along a Hilbert curve (None below `min_points`, where
    ordering costs more than it saves).  Spatially coherent waves are what lets the mesh kernels skip far tiles and the
    bucketed composed kernel skip far leaves.  One C-ABI call (pvamd_morton_order -- the name is older than the curve: a
    seven-launch counting sort on the cells of the curve); nothing comes back to the host.  With want_inverse / want_sorted: (order, inverse, sorted points)."""
    P = points.shape[0]
    if P < min_points or P == 0:
        return (None, None, None) if (want_inverse or want_sorted) else None
    dev = points.device
    order = torch.empty((P,), dtype=torch.int32, device=dev)
    inv = torch.empty((P,), dtype=torch.int32, device=dev) if want_inverse else None
    spts = torch.empty((P, 3), dtype=torch.float32, device=dev) if want_sorted else None
    scratch = torch.empty((morton_order_scratch_words(P),), dtype=torch.int32, device=dev)
    check(load().pvamd_morton_order(ptr(points), P, ptr(order), ptr(inv), ptr(spts), ptr(scratch), stream_ptr()),
          "pvamd_morton_order")
    return (order, inv, spts) if (want_inverse or want_sorted) else order
