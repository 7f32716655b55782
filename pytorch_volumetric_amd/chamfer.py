"""Unidirectional points -> object chamfer distance and the pose-set metrics built on it
(reference chamfer.py:12-195).  `batch_chamfer_dist` is one fused kernel per call (transform, distance, squared
reduction); the callers stay thin Python like the reference's."""
from typing import NamedTuple

import ctypes
import torch

from pytorch_volumetric_amd import _lib
from pytorch_volumetric_amd import transforms as tf
from pytorch_volumetric_amd.sdf import ObjectFactory, ObjectFrameSDF, CachedSDF, OutOfBoundsStrategy, \
    sample_mesh_points


def _matrix_to_rotation_6d(r):
    return r[..., :2, :].clone().reshape(*r.shape[:-2], 6)


def pairwise_distance(world_to_link_tfs):
    """cdist over (translation, 6-D rotation) pose vectors (chamfer.py:12-17)."""
    m = tf.as_matrix(world_to_link_tfs)
    cont_rep = torch.cat((m[:, :3, 3], _matrix_to_rotation_6d(m[:, :3, :3])), dim=1)
    return torch.cdist(cont_rep, cont_rep)


FLAT_MAX_POINTS_PER_TRANSFORM = 16384  # above this a transform's own points fill its waves with neighbours
FLAT_MAX_TOTAL_POINTS = 1 << 26        # the transformed copy is B * N * 12 bytes


def flat_call_pays(B, N):
    """Whether batch_chamfer_dist against a mesh transforms all B x N points first and queries them in one spatial order
    (pvamd_chamfer_mesh_flat) instead of launching per (transform, 64 points of that transform)."""
    return B >= 2 and 0 < N <= FLAT_MAX_POINTS_PER_TRANSFORM and B * N <= FLAT_MAX_TOTAL_POINTS


def batch_chamfer_dist(world_to_object: torch.tensor, model_points_world_frame_eval: torch.tensor,
                       obj_factory: ObjectFactory = None, obj_sdf: ObjectFrameSDF = None, viewing_delay=0, scale=1000.,
                       print_err=False, vis=None, reduce_group=None):
    """
    Batched unidirectional chamfer distance between world-frame surface points and an object under B candidate
    world->object transforms (chamfer.py:62-94).

    :param world_to_object: B x 4 x 4 transforms from world to object frame
    :param model_points_world_frame_eval: N x 3 points
    :param obj_factory: object (mesh) to evaluate against
    :param obj_sdf: SDF of the object to evaluate against (faster, less accurate); takes precedence like the reference
    :param scale: unit conversion applied to the distance before squaring (1000: m -> mm)
    :param reduce_group: optional torch.distributed group over which the N points are sharded: partial sums are
        all-reduced and divided by the global N
    :return: B chamfer errors, mean over the N points of (scale * d)^2
    """
    W = tf.as_matrix(world_to_object)
    out_dtype, out_device = W.dtype, W.device
    if obj_sdf is None and obj_factory is None:
        raise ValueError("Either obj_sdf or obj_factory must be given")
    sums, total_n = chamfer_partial_sums(W, model_points_world_frame_eval, obj_factory, obj_sdf, scale)
    if reduce_group is not None:
        import torch.distributed as dist
        count = torch.tensor([float(total_n)], dtype=torch.float64, device=sums.device)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=reduce_group)
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=reduce_group)
        total_n = count.item()
    return (sums / total_n).to(device=out_device, dtype=out_dtype)  # no points at all: 0 / 0 = NaN, torch's mean of nothing


def chamfer_partial_sums(W, points, obj_factory, obj_sdf, scale):
    """The per-rank half of batch_chamfer_dist: sum over THESE points of (scale * d)^2 per transform, float64 on the GPU, and
    the number of points -- what a run sharded over the points all-reduces (dist.sharded_chamfer).  W: (B, 4, 4)."""
    B = W.shape[0]
    lib = _lib.load()
    dev = _lib.require_gpu()
    pts = torch.as_tensor(points).detach().reshape(-1, 3).to(device=dev, dtype=torch.float32)
    pts = pts.contiguous()
    N = pts.shape[0]
    Wd = W.detach().to(device=dev, dtype=torch.float32).contiguous()
    sums = torch.empty((B,), dtype=torch.float64, device=dev)
    if N == 0:  # a rank whose shard is empty (fewer points than ranks): nothing to launch, it adds 0 to the all-reduce
        return sums.zero_(), 0

    fused_grid = isinstance(obj_sdf, CachedSDF) and obj_sdf._dim == 3 and \
        obj_sdf.out_of_bounds_strategy == OutOfBoundsStrategy.BOUNDING_BOX
    with _lib.on_device(dev):
        if fused_grid:
            desc = obj_sdf._grid_desc()
            _lib.check(lib.pvamd_chamfer_grid(ctypes.byref(desc), _lib.ptr(Wd), B, _lib.ptr(pts), N, float(scale),
                                              _lib.ptr(sums), _lib.stream_ptr()), "pvamd_chamfer_grid")
        elif obj_sdf is not None:
            # arbitrary SDF object: transform on device, query it, reduce
            x = pts.unsqueeze(0) @ Wd[:, :3, :3].transpose(-1, -2) + Wd[:, None, :3, 3]
            d, _ = obj_sdf(x)
            sums = ((float(scale) * d.to(device=dev, dtype=torch.float32)) ** 2).double().sum(dim=-1)
        else:
            desc = obj_factory._mesh_desc()
            if flat_call_pays(B, N):
                # many transforms x few points (pairwise_distance_chamfer / PlausibleDiversity: 10^4 x 500): transform all
                # B * N points once, walk them in ONE spatial order (the 64 points of a wave are then neighbours in space,
                # whichever transform they came from) and let every point add to its own transform's sum
                x = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
                for b0 in range(0, B, 65535):  # the transform kernel carries the transform in a grid dimension
                    nb = min(65535, B - b0)
                    _lib.check(lib.pvamd_transform_points(_lib.ptr(Wd[b0:b0 + nb]), nb, _lib.ptr(pts), N, _lib.ptr(x[b0:b0 + nb]),
                                                          _lib.stream_ptr()), "pvamd_transform_points")
                flat = x.view(-1, 3)
                order = _lib.morton_order(flat)
                scratch = torch.empty((_lib.mesh_scratch_bytes(B * N) // 8,), dtype=torch.int64, device=dev)
                _lib.check(lib.pvamd_chamfer_mesh_flat(ctypes.byref(desc), B, _lib.ptr(flat), _lib.ptr(order), N, float(scale),
                                                       _lib.ptr(sums), _lib.ptr(scratch), _lib.stream_ptr()),
                           "pvamd_chamfer_mesh_flat")
            else:
                order = _lib.morton_order(pts)
                # slots for the point groups the kernel hands over (those about equidistant to much of the mesh)
                scratch = torch.empty((_lib.mesh_scratch_bytes(N) // 8,), dtype=torch.int64, device=dev)
                _lib.check(lib.pvamd_chamfer_mesh(ctypes.byref(desc), _lib.ptr(Wd), B, _lib.ptr(pts), _lib.ptr(order), N,
                                                  float(scale), _lib.ptr(sums), _lib.ptr(scratch), _lib.stream_ptr()),
                           "pvamd_chamfer_mesh")
    return sums, N


def pairwise_distance_chamfer(A_link_to_world_tfs, B_world_to_link_tfs=None,
                              obj_factory: ObjectFactory = None, obj_sdf: ObjectFrameSDF = None,
                              model_points_eval: torch.tensor = None, vis=None, scale=1000):
    """B x P chamfer matrix between two sets of poses of one object (chamfer.py:20-59)."""
    T = tf.as_matrix(A_link_to_world_tfs)
    if model_points_eval is None:
        model_points_eval, _, _ = sample_mesh_points(obj_factory, num_points=500, name=obj_factory.name,
                                                     device=T.device)
    T_inv = tf.rigid_inverse(T) if B_world_to_link_tfs is None else tf.as_matrix(B_world_to_link_tfs)
    Iapprox = torch.einsum("bij,pjk->bpik", T_inv, T)
    B, P = len(T), len(T_inv)
    errors = batch_chamfer_dist(Iapprox.reshape(B * P, 4, 4), model_points_eval, obj_factory=obj_factory,
                                obj_sdf=obj_sdf, viewing_delay=0, vis=vis, scale=scale)
    return errors.view(B, P)


class MinOverAxis(NamedTuple):
    """What torch.min(dim=...) returns: the reference hands these out as most_plausible_per_estimated / most_covered_per_plausible."""
    values: torch.Tensor
    indices: torch.Tensor


class PlausibleDiversityReturn(NamedTuple):
    plausibility: torch.tensor
    coverage: torch.tensor
    most_plausible_per_estimated: MinOverAxis
    most_covered_per_plausible: MinOverAxis


def reduce_pairwise_errors(errors):
    """(B, P) chamfer matrix -> PlausibleDiversityReturn (chamfer.py:185-195): plausibility = mean over the B estimated poses
    of their best match among the P plausible ones, coverage = mean over the P plausible poses of their best match among
    the estimated ones, with the arg-minima.  A matrix on the GPU is reduced by ONE pass of pvamd_pairwise_min_reduce (row and
    column minima + indices, then the two means in float64 in a fixed order); a host tensor by the two torch reductions."""
    if errors.dim() != 2:
        raise ValueError(f"expected a (B, P) error matrix, got {tuple(errors.shape)}")
    B, P = errors.shape
    if not (errors.is_cuda and errors.dtype in (torch.float32, torch.float64) and B > 0 and P > 0):
        rows, cols = errors.min(dim=1), errors.min(dim=0)
        return PlausibleDiversityReturn(rows.values.sum() / B, cols.values.sum() / P, MinOverAxis(rows.values, rows.indices),
                                        MinOverAxis(cols.values, cols.indices))
    E = errors.detach().contiguous()
    dev = E.device
    row_val, col_val = torch.empty((B,), dtype=E.dtype, device=dev), torch.empty((P,), dtype=E.dtype, device=dev)
    row_idx, col_idx = torch.empty((B,), dtype=torch.int64, device=dev), torch.empty((P,), dtype=torch.int64, device=dev)
    means = torch.empty((2,), dtype=torch.float64, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.load().pvamd_pairwise_min_reduce(_lib.ptr(E), 1 if E.dtype == torch.float64 else 0, B, P, _lib.ptr(row_val),
                                                         _lib.ptr(row_idx), _lib.ptr(col_val), _lib.ptr(col_idx), _lib.ptr(means),
                                                         _lib.stream_ptr()), "pvamd_pairwise_min_reduce")
    means = means.to(E.dtype)
    return PlausibleDiversityReturn(means[0], means[1], MinOverAxis(row_val, row_idx), MinOverAxis(col_val, col_idx))


class PlausibleDiversity:
    """Plausibility / coverage of an estimated pose set against a plausible pose set, in squared distance units (chamfer.py:
    123-195): every (estimated, plausible) pair is scored by the chamfer error of the object's surface points under the
    relative pose, and the (B, P) matrix is reduced on the device."""

    def __init__(self, obj_factory: ObjectFactory, model_points_eval: torch.tensor = None, num_model_points_eval=500,
                 obj_sdf: ObjectFrameSDF = None):
        self.obj_factory, self.obj_sdf = obj_factory, obj_sdf
        if model_points_eval is None:  # the object's own surface samples (chamfer.py:136-138)
            model_points_eval = sample_mesh_points(obj_factory, num_points=num_model_points_eval, name=obj_factory.name)[0]
        self.model_points_eval = model_points_eval

    def pairwise_errors(self, first_inv, second, scale=1000.):
        """errors[b, p] = chamfer error of the model points under first_inv[b] @ second[p] (chamfer.py:173-183).  All B * P
        relative poses go through ONE batch_chamfer_dist call (one spatial order over every transformed point)."""
        relative = torch.matmul(tf.as_matrix(first_inv).unsqueeze(1), tf.as_matrix(second).unsqueeze(0))  # (B, P, 4, 4)
        B, P = relative.shape[:2]
        self.model_points_eval = self.model_points_eval.to(device=relative.device, dtype=relative.dtype)
        return batch_chamfer_dist(relative.reshape(B * P, 4, 4), self.model_points_eval, self.obj_factory, obj_sdf=self.obj_sdf,
                                  scale=scale).view(B, P)

    # the reference's names for the two steps, with its parameter names (chamfer.py:173, 185)
    def compute_tf_pairwise_error_per_batch(self, T_est_inv, T_p, scale=1000.):
        return self.pairwise_errors(T_est_inv, T_p, scale=scale)

    @staticmethod
    def do_evaluate_plausible_diversity_on_pairwise_chamfer_dist(errors_per_batch):
        return reduce_pairwise_errors(errors_per_batch)

    def __call__(self, T_est_inv, T_p, bidirectional=False, scale=1000.):
        forward = reduce_pairwise_errors(self.pairwise_errors(T_est_inv, T_p, scale=scale))
        if not bidirectional:
            return forward
        # the sets swapped: what was plausibility there is coverage here and vice versa (chamfer.py:160-170); the argmins
        # reported stay those of the forward direction
        backward = reduce_pairwise_errors(self.pairwise_errors(T_p, T_est_inv, scale=scale))
        return PlausibleDiversityReturn((forward.plausibility + backward.coverage) / 2, (forward.coverage + backward.plausibility) / 2,
                                        forward.most_plausible_per_estimated, forward.most_covered_per_plausible)
