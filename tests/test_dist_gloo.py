"""CPU, world_size 2 over gloo: the multi-GPU sharding logic (partition, pad, all-gather, reorder, trim; chamfer
all-reduce) with the oracle standing in for the per-rank kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pytorch_volumetric_amd as pv
from oracle import oracle
from pytorch_volumetric_amd.dist import ShardedSDF, shard_range
from tests import helpers as H


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleLeaf:
    """CPU stand-in for a CachedSDF leaf (the checker computes here; the product path would be the HIP kernel)."""

    def __init__(self):
        gt = H.drill_like_gt()
        rng = H.padded_range(H.DRILL_BB, 0.1)
        snapped = pv.get_divisible_range_by_resolution(0.02, rng)
        _, pts = pv.get_coordinates_and_points_in_grid(0.02, snapped)
        coords, _ = pv.get_coordinates_and_points_in_grid(0.02, snapped)
        v, g = gt(pts)
        self.grid = oracle.Grid(v.reshape([len(c) for c in coords]).numpy(), g.numpy(),
                                np.array([s[0] for s in snapped]), np.array([s[1] for s in snapped]), H.DRILL_BB)

    def __call__(self, pts):
        lead = pts.shape[:-1]
        v, g, _ = oracle.cached_query(self.grid, pts.reshape(-1, 3).numpy())
        return torch.from_numpy(v).reshape(*lead), torch.from_numpy(g).reshape(*lead, 3)


class OracleRobot:
    """Config-batched composed query: returns (A, P) / (A, P, 3) like RobotSDF."""

    def __init__(self, A):
        self.leaf = OracleLeaf()
        self.A = A
        self.tf = H.random_rigid(3 * A, seed=11).numpy()

    def __call__(self, pts):
        v, g, _ = oracle.composed_query([self.leaf.grid] * 3, self.tf, self.A, pts.reshape(-1, 3).numpy())
        return torch.from_numpy(v), torch.from_numpy(g)


class OracleFlatComposed:
    """A composition WITHOUT a transform batch: like the reference (sdf.py:418-426,433) it returns FLAT (P,) / (P,3) even
    for batched points.  Carries the two attributes ShardedSDF recognises such an object by."""

    def __init__(self):
        self.leaf = OracleLeaf()
        self.sdfs = [self.leaf] * 3
        self.tsf_batch = None
        self.tf = H.random_rigid(3, seed=12).numpy()

    def __call__(self, pts):
        v, g, _ = oracle.composed_query([self.leaf.grid] * 3, self.tf, 1, pts.reshape(-1, 3).numpy())
        return torch.from_numpy(v[0]), torch.from_numpy(g[0])


def worker(rank, world, port, P, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pts = H.uniform_points(P, [-0.3] * 3, [0.4] * 3, seed=5).reshape(-1, 3)
        leaf = OracleLeaf()
        full_v, full_g = leaf(pts)
        sv, sg = ShardedSDF(leaf)(pts)
        ok = torch.equal(sv, full_v) and torch.equal(sg.nan_to_num(7.), full_g.nan_to_num(7.))
        # batched point dims come back in their original shape
        if P % 7 == 0:
            bv, bg = ShardedSDF(leaf)(pts.reshape(7, P // 7, 3))
            ok = ok and bv.shape == (7, P // 7) and torch.equal(bv.reshape(-1), full_v)
        # configuration batch: (A, P) outputs gathered along the point axis
        robot = OracleRobot(A=4)
        rv, rg = robot(pts)
        srv, srg = ShardedSDF(robot)(pts)
        ok = ok and srv.shape == (4, P) and torch.equal(srv, rv) and torch.equal(srg.nan_to_num(7.), rg.nan_to_num(7.))
        # a composition without a transform batch stays flat for batched points, sharded or not
        if P % 7 == 0:
            flat = OracleFlatComposed()
            fv, fg = flat(pts.reshape(7, P // 7, 3))
            sfv, sfg = ShardedSDF(flat)(pts.reshape(7, P // 7, 3))
            ok = ok and fv.shape == (P,) and sfv.shape == (P,) and sfg.shape == (P, 3) and torch.equal(sfv, fv)
        # gather=False: the local slice only
        lv, lg, (a, b) = ShardedSDF(leaf, gather=False)(pts)
        ok = ok and (a, b) == shard_range(P, world, rank)[:2] and torch.equal(lv, full_v[a:b])
        # chamfer partial sums: all-reduce of B float64 + the global point count
        sums = torch.tensor([float(rank + 1), 2.0], dtype=torch.float64)
        dist.all_reduce(sums)
        ok = ok and sums.tolist() == [3.0, 4.0]
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P", [1001, 14, 1])
def test_sharded_query_equals_unsharded_world2(P):
    world = 2
    port = free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(worker, args=(world, port, P, results), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


def test_shard_range_covers_everything_exactly_once():
    for P in (0, 1, 7, 8, 1000, 1001):
        for W in (1, 2, 3, 8):
            spans = [shard_range(P, W, r) for r in range(W)]
            covered = sum(b - a for a, b, _ in spans)
            assert covered == P
            assert all(spans[i][1] == spans[i + 1][0] or spans[i + 1][0] == P for i in range(W - 1))
            assert all(b - a <= c for a, b, c in spans)


class OraclePackedRobot(OracleRobot):
    """Stand-in for a fused RobotSDF: what ShardedSDF's ONE-collective path calls (query_packed = packed (val, grad) records
    of this rank's 256-padded slice) and what its configuration sharding calls (query_configs), computed by the oracle."""

    def __init__(self, A, batch):
        super().__init__(A)
        self.tsf_batch = batch
        self.sdfs = [type("Leaf", (), {"device": "cpu"})()] * 3

    def _fusable(self):
        return True

    def _owner_device(self):
        return torch.device("cpu")

    def query_packed(self, pts):
        assert pts.shape[0] % 256 == 0
        v, g = self(pts)
        return torch.cat((v.unsqueeze(-1), g), dim=-1).contiguous()  # (A, Pp, 4)

    def query_configs(self, pts, first, count):
        pick = np.minimum(np.arange(first, first + count), self.A - 1)
        tf = self.tf.reshape(3, self.A, 4, 4)[:, pick].reshape(-1, 4, 4)
        v, g, _ = oracle.composed_query([self.leaf.grid] * 3, tf, count, pts.reshape(-1, 3).numpy())
        return torch.from_numpy(v), torch.from_numpy(g)


class CpuUnpack(ShardedSDF):
    """ShardedSDF with the unpack KERNEL replaced by the statement it implements (out[a][j] = rec[a * Pp + index[j]]): the
    partition / padding / collective / index logic under test is the product's own."""

    def _unpack_records(self, gathered, index, P, Pp, A, dev):
        rec = gathered.reshape(-1, 4)
        rows = (torch.arange(A).unsqueeze(1) * Pp + index.long().unsqueeze(0))  # (A, P)
        out = rec[rows]
        return out[..., 0].contiguous(), out[..., 1:].contiguous()


def packed_worker(rank, world, port, P, A, batch, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pts = H.uniform_points(P, [-0.3] * 3, [0.4] * 3, seed=9)
        robot = OraclePackedRobot(A, batch)
        rv, rg = robot(pts)
        sh = CpuUnpack(robot)
        sv, sg = sh(pts)
        ok = sh.last_path == "packed" and sv.shape == (*batch, P) and sg.shape == (*batch, P, 3)
        ok = ok and torch.equal(sv.reshape(A, P), rv) and torch.equal(sg.reshape(A, P, 3).nan_to_num(7.), rg.nan_to_num(7.))
        chunk = -(-P // world)
        ok = ok and sh.bytes_received_per_rank == (world - 1) * A * (-(-chunk // 256) * 256) * 16
        # batched point dims
        if P % 4 == 0:
            bv, _ = sh(pts.reshape(4, P // 4, 3))
            ok = ok and bv.shape == (*batch, 4, P // 4) and torch.equal(bv.reshape(A, P), rv)
        # sharded over configurations instead: rows gathered in place, odd A pads the last rank's share
        sc = ShardedSDF(robot, shard="configs")
        cv, cg = sc(pts)
        ok = ok and sc.last_path == "configs" and cv.shape == (*batch, P)
        ok = ok and torch.equal(cv.reshape(A, P), rv) and torch.equal(cg.reshape(A, P, 3).nan_to_num(7.), rg.nan_to_num(7.))
        lv, lg, (a0, a1) = ShardedSDF(robot, shard="configs", gather=False)(pts)
        ok = ok and (a0, a1) == shard_range(A, world, rank)[:2] and torch.equal(lv, rv[a0:a1]) and lg.shape == (a1 - a0, P, 3)
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P,A,batch", [(1001, 5, (5,)), (512, 6, (2, 3)), (3, 1, (1,))])
def test_packed_one_collective_path_and_config_sharding_world2(P, A, batch):
    """The packed path's index / padding / unpack bookkeeping and shard='configs', over gloo with the oracle standing in
    for the kernels (round 2 covered the packed path only on the GPU box)."""
    world = 2
    port = free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(packed_worker, args=(world, port, P, A, batch, results), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


def test_config_sharding_needs_a_batch():
    with pytest.raises(ValueError):
        ShardedSDF(OracleLeaf(), shard="rows")


def chamfer_worker(rank, world, port, N, results):
    """sharded_chamfer with the kernel half (chamfer.chamfer_partial_sums) replaced by the oracle on the CPU: what is under
    test is the split, the empty shard, the all-reduce of B float64 sums + the count and the division by the GLOBAL N."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_volumetric_amd import chamfer
        leaf = OracleLeaf()
        seen = []

        def partial_sums_on_the_cpu(W, points, obj_factory, obj_sdf, scale):
            pts = torch.as_tensor(points).reshape(-1, 3)
            seen.append(pts.shape[0])
            sums = oracle.chamfer_grid(obj_sdf.grid, W.numpy(), pts.numpy(), scale) if len(pts) else np.zeros(len(W))
            return torch.from_numpy(sums), pts.shape[0]

        chamfer.chamfer_partial_sums = partial_sums_on_the_cpu
        B = 3
        W = H.random_rigid(B, seed=21, trans=0.05)
        pts = H.uniform_points(N, [-0.1] * 3, [0.2] * 3, seed=22)
        got = pv.sharded_chamfer(W, pts, obj_sdf=leaf, scale=1000.0)
        whole = oracle.chamfer_grid(leaf.grid, W.numpy(), pts.numpy(), 1000.0) / N
        start, stop, _ = shard_range(N, world, rank)
        ok = seen == [stop - start] and got.shape == (B,) and got.dtype == W.dtype
        # float64 partial sums in a different order than the single loop: 1e-12 relative, not bit equality
        ok = ok and np.allclose(got.double().numpy(), whole, rtol=1e-6 if got.dtype == torch.float32 else 1e-12, atol=0)
        results[rank] = (bool(ok), stop - start)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,world", [(1001, 2), (1, 2), (2, 3), (7, 3)])
def test_sharded_chamfer_unequal_and_empty_shards(N, world):
    """an odd point count (unequal shards) and fewer points than ranks (a rank whose shard is empty adds
    zero sums and a zero count): the mean is over the global N either way (chamfer.py:79-94)."""
    port = free_port()
    results = mp.Manager().dict()
    mp.spawn(chamfer_worker, args=(world, port, N, results), nprocs=world, join=True)
    got = dict(results)
    assert all(ok for ok, _ in got.values()), got
    assert sum(n for _, n in got.values()) == N
    if N < world:
        assert any(n == 0 for _, n in got.values())
