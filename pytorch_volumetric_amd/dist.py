"""Multi-GPU sharding of SDF queries: one process per GPU, query points split across ranks, one all-gather
(RCCL over xGMI when the backend is "nccl") to reassemble (sdf_val, sdf_grad).  For a composition of cached grids
(RobotSDF / ComposedSDF, the C4 case) the kernel writes packed (val, grad) records straight into the send buffer, ONE
all-gather moves them, and one unpack kernel writes the final (A, P) / (A, P, 3) layout: no second collective and no
strided copy of the gathered tensors.

Every query point is independent (reference sdf.py:535-591, 392-433, 122-172 have no cross-point term), the
read-only state (voxel grids, meshes, transforms) is small and replicated on every GPU, so the only communication
is the gather of the outputs -- or none at all with gather=False, when the consumer can use sharded results.
"""
import math

import torch
import torch.distributed as dist

from pytorch_volumetric_amd import _lib


def shard_range(num_points, world_size, rank):
    """Rank r owns [r*chunk, min(P, (r+1)*chunk)) with chunk = ceil(P / W); returns (start, stop, chunk)."""
    chunk = (num_points + world_size - 1) // world_size if num_points > 0 else 0
    start = min(num_points, rank * chunk)
    stop = min(num_points, start + chunk)
    return start, stop, chunk


_INTO_TENSOR = {}


def _gather_into(out, local, group):
    """All-gather `local` into `out` (W, *local.shape).  Backends with all_gather_into_tensor (nccl = RCCL) get the one
    flat collective; the others (gloo, the CPU tests) the list form.  Which one is decided ONCE per backend, from the
    backend's name and -- for a backend this module has not seen -- a first attempt; a failure of the collective itself
    is never answered with a second collective."""
    backend = dist.get_backend(group)
    into = _INTO_TENSOR.get(backend)
    if into is None and backend == "gloo":
        into = _INTO_TENSOR[backend] = False
    if into is None:  # probe: NotImplementedError / "not supported" RuntimeError are raised before anything is sent
        try:
            dist.all_gather_into_tensor(out, local, group=group)
            _INTO_TENSOR[backend] = True
            return
        except NotImplementedError:
            into = _INTO_TENSOR[backend] = False
        except RuntimeError as e:
            if "support" not in str(e).lower() and "implement" not in str(e).lower():
                raise
            into = _INTO_TENSOR[backend] = False
    if into:
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        dist.all_gather([out[r] for r in range(out.shape[0])], local, group=group)


def _all_gather_cat(local, point_dim, group):
    """Gather equal-shaped `local` tensors from all ranks and concatenate them along `point_dim`."""
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    _gather_into(out, local.contiguous(), group)
    # (W, ..., chunk, ...) -> (..., W*chunk, ...): the strided copy that restores the global point order
    point_dim = point_dim % local.dim()
    out = out.movedim(0, point_dim)  # (..., W, chunk, ...)
    shape = list(local.shape)
    shape[point_dim] = world * local.shape[point_dim]
    return out.reshape(shape)


def packed_index(P, chunk, A, Pp, device):
    """Where caller point j sits in the gathered buffer of packed records, laid out (world, A, Pp, 4): in the slab of rank
    j // chunk, at position j % chunk of configuration 0 -- the unpack adds a * Pp for configuration a."""
    j = torch.arange(P, device=device, dtype=torch.int64)
    return ((j // chunk) * (A * Pp) + j % chunk).to(torch.int32)


class ShardedSDF:
    """Wrap any ObjectFrameSDF so that __call__ evaluates only this rank's share of the query and all-gathers the results.

    shard="points" (BASELINE.json's split): this rank's slice of the (flattened) query points, all configurations.
    shard="configs" (RobotSDF / batched ComposedSDF only): every point, this rank's slice of the configuration batch --
    the rows of (A, P) / (A, P, 3) a rank produces are contiguous in the result, so the all-gather writes them in place:
    no packed records, no unpack pass; worthwhile when A >= the number of ranks.

    Output shapes match the wrapped SDF's for the full input: leading configuration dims (RobotSDF / batched
    ComposedSDF) are preserved, the point dims come back in the original order.
    """

    def __init__(self, sdf, group=None, gather=True, compute_device=None, shard="points"):
        if shard not in ("points", "configs"):
            raise ValueError(f"shard must be 'points' or 'configs', got {shard!r}")
        self.sdf = sdf
        self.group = group
        self.gather = gather
        self.compute_device = compute_device  # where gathered tensors live (nccl needs GPU tensors)
        self.shard = shard

    def surface_bounding_box(self, **kwargs):
        return self.sdf.surface_bounding_box(**kwargs)

    def __call__(self, points_in_object_frame):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        lead = tuple(points_in_object_frame.shape[:-1])
        self._query_dtype = points_in_object_frame.dtype if points_in_object_frame.dtype.is_floating_point else torch.float32
        flat = points_in_object_frame.reshape(-1, 3)
        P = flat.shape[0]
        if self.shard == "configs":
            return self._configs_call(flat, lead, world, rank)
        start, stop, chunk = shard_range(P, world, rank)
        packed = self._packed_call(flat, lead, world, start, stop, chunk)
        if packed is not None:
            return packed
        mine = flat[start:stop]
        if mine.shape[0] < chunk:  # pad the tail so every rank contributes the same count (all-gather needs it)
            pad = flat[:1].expand(chunk - mine.shape[0], 3) if P > 0 else flat.new_zeros((chunk, 3))
            mine = torch.cat((mine, pad), dim=0)
        self.last_path = "generic"
        val, grad = self._query(mine, start)
        # leaf / composed-without-batch: val (chunk,), grad (chunk,3); with a configuration batch: (A..., chunk[,3])
        if not self.gather:
            n_valid = stop - start
            return val[..., :n_valid], grad[..., :n_valid, :], (start, stop)
        if self.compute_device is not None:
            val, grad = val.to(self.compute_device), grad.to(self.compute_device)
        val = _all_gather_cat(val, -1, self.group)[..., :P]
        grad = _all_gather_cat(grad, -2, self.group)[..., :P, :]
        batch = tuple(val.shape[:-1])
        if self._returns_flat():
            return val.reshape(*batch, -1), grad.reshape(*batch, -1, 3)
        return val.reshape(*batch, *lead), grad.reshape(*batch, *lead, 3)

    def _configs_call(self, flat, lead, world, rank):
        """shard="configs": rank r evaluates configurations [r * per, (r + 1) * per) of the flattened batch at every point
        (per = ceil(A / W); the last rank repeats configuration A - 1 to fill its share, trimmed after the gather)."""
        inner = getattr(self.sdf, "sdf", self.sdf)  # RobotSDF -> its ComposedSDF
        batch = getattr(inner, "tsf_batch", None)
        if batch is None or not hasattr(inner, "query_configs"):
            raise ValueError("shard='configs' needs a RobotSDF / ComposedSDF with a configuration batch")
        A = math.prod(batch)
        start, stop, per = shard_range(A, world, rank)
        self.last_path = "configs"
        val, grad = inner.query_configs(flat, start, per)  # (per, P), (per, P, 3); float64 for float64 points
        if val.dtype != self._query_dtype:  # e.g. float16 points: results in the query dtype, as ComposedSDF.__call__ returns them
            val, grad = val.to(self._query_dtype), grad.to(self._query_dtype)
        if not self.gather:
            return val[:stop - start], grad[:stop - start], (start, stop)
        if self.compute_device is not None:
            val, grad = val.to(self.compute_device), grad.to(self.compute_device)
        P = flat.shape[0]
        out_val = torch.empty((world, per, P), dtype=val.dtype, device=val.device)
        out_grad = torch.empty((world, per, P, 3), dtype=grad.dtype, device=grad.device)
        _gather_into(out_val, val.contiguous(), self.group)   # rows land where (A, P) wants them: no reorder pass
        _gather_into(out_grad, grad.contiguous(), self.group)
        self.bytes_received_per_rank = (world - 1) * per * P * 16
        val, grad = out_val.reshape(world * per, P)[:A], out_grad.reshape(world * per, P, 3)[:A]
        return val.reshape(*batch, *lead), grad.reshape(*batch, *lead, 3)

    def _unpack_records(self, gathered, index, P, Pp, A, dev):
        """(world, A, Pp, 4) packed records -> val (A, P), grad (A, P, 3) in caller point order: one kernel."""
        val = torch.empty((A, P), dtype=torch.float32, device=dev)
        grad = torch.empty((A, P, 3), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.check(_lib.load().pvamd_unpack_records(_lib.ptr(gathered), _lib.ptr(index), P, Pp, A, _lib.ptr(val),
                                                        _lib.ptr(grad), _lib.stream_ptr()), "pvamd_unpack_records")
        return val, grad

    def _packed_call(self, flat, lead, world, start, stop, chunk):
        """The one-collective path (module docstring); None when it does not apply."""
        inner = getattr(self.sdf, "sdf", self.sdf)  # RobotSDF -> its ComposedSDF
        P = flat.shape[0]
        if not (self.gather and P > 0 and hasattr(inner, "query_packed") and inner._fusable() and torch.is_tensor(flat)
                and flat.dtype == torch.float32):
            return None
        dev = inner._owner_device()
        if self.compute_device is not None and torch.device(self.compute_device).type != "cuda":
            return None
        A = math.prod(inner.tsf_batch) if inner.tsf_batch is not None else 1
        Pp = -(-chunk // 256) * 256
        if world * A * Pp >= 2 ** 31:  # the unpack kernel indexes records with int32
            return None
        self.last_path = "packed"
        mine = flat[start:stop].to(dev)
        if mine.shape[0] < Pp:  # whole 256-point tiles: pad with copies of a point (their records are never unpacked)
            mine = torch.cat((mine, flat[:1].to(dev).expand(Pp - mine.shape[0], 3)), dim=0)
        rec = inner.query_packed(mine.contiguous())
        gathered = torch.empty((world, A, Pp, 4), dtype=torch.float32, device=dev)
        _gather_into(gathered, rec, self.group)
        self.bytes_received_per_rank = (world - 1) * A * Pp * 16
        key = (P, chunk, A, Pp, str(dev))
        if getattr(self, "_index_key", None) != key:
            self._index = packed_index(P, chunk, A, Pp, dev)
            self._index_key = key
        val, grad = self._unpack_records(gathered, self._index, P, Pp, A, dev)
        # the same device / dtype convention as ComposedSDF.__call__ (whatever path ran): results on the leaves' own
        # device (sdf.py:546), in the dtype of the query points
        out_device, out_dtype = inner.sdfs[0].device, self._query_dtype
        if self.compute_device is not None:
            out_device = self.compute_device
        val, grad = val.to(device=out_device, dtype=out_dtype), grad.to(device=out_device, dtype=out_dtype)
        if inner.tsf_batch is None:
            return val.reshape(-1), grad.reshape(-1, 3)
        return val.reshape(*inner.tsf_batch, *lead), grad.reshape(*inner.tsf_batch, *lead, 3)

    def _returns_flat(self):
        """A ComposedSDF (or RobotSDF) WITHOUT a transform batch returns flat (P,) / (P,3) even for batched points
        (reference sdf.py:418-426,433); the wrapper keeps the wrapped SDF's own convention."""
        inner = getattr(self.sdf, "sdf", self.sdf)  # RobotSDF -> its ComposedSDF
        return hasattr(inner, "tsf_batch") and hasattr(inner, "sdfs") and inner.tsf_batch is None

    def _query(self, pts, index_base):
        # MeshSDF's sign jitter is indexed by the global point number so sharding does not change the result
        factory = getattr(self.sdf, "obj_factory", None)
        if factory is not None and hasattr(factory, "object_frame_closest_point"):
            res = factory.object_frame_closest_point(pts, index_base=index_base)
            return res.distance, res.gradient
        return self.sdf(pts)


def sharded_chamfer(world_to_object, points, obj_factory=None, obj_sdf=None, scale=1000., group=None):
    """batch_chamfer_dist with the N source points sharded across ranks: each rank reduces its slice, the B partial
    sums are all-reduced (a few KB) and divided by the global N."""
    from pytorch_volumetric_amd.chamfer import batch_chamfer_dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    start, stop, _ = shard_range(points.shape[0], world, rank)
    return batch_chamfer_dist(world_to_object, points[start:stop], obj_factory=obj_factory, obj_sdf=obj_sdf,
                              scale=scale, reduce_group=group if group is not None else dist.group.WORLD)
