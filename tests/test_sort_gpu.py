"""-m gpu: pvamd_morton_order -- the seven-launch counting sort and, from 786,432 points (3 << 18), the hand-written LSD radix sort -- that
gives the mesh kernels and the bucketed composed path their spatial processing order (replaces torch.argsort of Morton keys,
VERDICT r1 item 8; no library sort since round 5)."""
import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def curve_cells(pts, bits):
    """Host restatement of the cell a point falls in and of its position along the Hilbert curve (csrc/morton.h
    hilbert_key30: 2^(bits/3) cells per axis over the bounds of the finite coordinates, Skilling's transform)."""
    b = bits // 3
    p = pts.astype(np.float32)
    fin = np.where(np.isfinite(p), p, np.nan)
    lo, hi = np.nanmin(fin, axis=0).astype(np.float32), np.nanmax(fin, axis=0).astype(np.float32)
    span = np.maximum((hi - lo).astype(np.float32), np.float32(1e-30))
    top = np.float32((1 << b) - 1)
    with np.errstate(invalid="ignore"):
        t = ((p - lo).astype(np.float32) * (np.float32(top + np.float32(0.999)) / span).astype(np.float32)).astype(np.float32)
    t = np.where(np.isnan(t), 0.0, np.clip(t, 0.0, top))
    X = [t[:, d].astype(np.int64) for d in range(3)]
    Q = 1 << (b - 1)
    while Q > 1:
        P = Q - 1
        for d in range(3):
            hit = (X[d] & Q) != 0
            t_ = np.where(hit, 0, (X[0] ^ X[d]) & P)
            X[0] = np.where(hit, X[0] ^ P, X[0] ^ t_)
            if d:
                X[d] = X[d] ^ t_
        Q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    g = np.zeros_like(X[0])
    Q = 1 << (b - 1)
    while Q > 1:
        g = np.where((X[2] & Q) != 0, g ^ (Q - 1), g)
        Q >>= 1
    key = np.zeros(len(p), dtype=np.int64)
    for k in range(b):
        for d in range(3):
            key |= (((X[d] ^ g) >> k) & 1) << (3 * k + (2 - d))
    return key


@pytest.mark.parametrize("P", [1, 5, 300, 10_000, 16_384, 16_385, 65_535, 65_536, 262_144, (1 << 20) + 77, (3 << 18) - 1, (3 << 18) + 77, (3 << 19) - 1,
                               (3 << 19) + 77, 1 << 21, (1 << 22) + 4095, 9_000_001])
def test_order_is_a_permutation_that_walks_the_cells_along_the_hilbert_curve(P):
    pts = H.uniform_points(P, [-0.7, -0.7, -0.2], [0.7, 0.7, 1.5], seed=P).cuda()
    order, inv, spts = _lib.morton_order(pts, min_points=0, want_inverse=True, want_sorted=True)
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(P))
    assert np.array_equal(inv.cpu().numpy()[o], np.arange(P))
    assert torch.equal(spts, pts[order.long()])
    cells = curve_cells(pts.cpu().numpy(), 21 if P >= (1 << 20) else (18 if P >= (1 << 16) else (15 if P > 16384 else 12)))
    walked = cells[o]
    assert (np.diff(walked) >= 0).all()  # cells in curve order; inside a cell any order ...
    if P >= (3 << 18):  # ... except from 786,432 points on (a stable radix sort of (cell, index) pairs, csrc/sort.hip): index order
        same = np.diff(walked) == 0
        assert (np.diff(o)[same] > 0).all()


def test_a_full_grid_is_walked_cell_to_face_neighbour():
    """One point per cell of a 32^3 grid: consecutive points of the order are face neighbours (the property the Z curve
    lacks and the 64-point groups of the mesh kernels are tighter for)."""
    n = 32
    idx = torch.stack(torch.meshgrid(*[torch.arange(n)] * 3, indexing="ij"), dim=-1).reshape(-1, 3).float()
    pts = ((idx + 0.5) / n)
    pts = torch.cat((pts, torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]])))  # pin the bounds to the grid's box
    order = _lib.morton_order(pts.cuda(), min_points=0).cpu().numpy()
    walked = order[order < n ** 3]
    steps = np.abs(np.diff(idx.numpy()[walked], axis=0)).sum(axis=1)
    assert (steps == 1).all()


def test_non_finite_points_are_placed_somewhere_and_nothing_else_moves():
    pts = H.uniform_points(5000, [-1] * 3, [1] * 3, seed=1)
    pts[7] = float("nan")
    pts[11, 1] = float("inf")
    pts[13, 2] = float("-inf")
    order = _lib.morton_order(pts.cuda(), min_points=0).cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(5000))


def test_degenerate_clouds():
    same = torch.ones(4096, 3).cuda() * 0.25  # zero extent: one cell
    assert np.array_equal(np.sort(_lib.morton_order(same, min_points=0).cpu().numpy()), np.arange(4096))
    line = torch.zeros(4096, 3)
    line[:, 0] = torch.linspace(0, 1, 4096)
    o = _lib.morton_order(line.cuda(), min_points=0).cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(4096))
    assert np.abs(np.diff(line[o, 0].numpy())).mean() < 2.0 / 15  # neighbours stay neighbours (16 cells per axis here)


def test_mesh_query_bits_do_not_depend_on_the_processing_order():
    """The order inside a cell is run-dependent; the mesh kernel's results are not."""
    obj = pv.MeshObjectFactory(H.mesh_path("probe.obj"))
    pts = H.uniform_points(30_000, [-0.05] * 3, [0.08] * 3, seed=2).cuda()
    a = obj.object_frame_closest_point(pts, compute_normal=True)
    b = obj.object_frame_closest_point(pts, compute_normal=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_radix_sort_is_deterministic_and_handles_clustered_keys():
    """From 786,432 points on: the same permutation on every run (stable passes, no atomics in the ranking), also when almost
    every point shares one cell (one digit run holds nearly the whole tile) and when the cloud is a few clusters."""
    P = (3 << 18) + 12_345
    g = torch.Generator().manual_seed(3)
    same = torch.full((P, 3), 0.5)
    same[:1000] = torch.rand(1000, 3, generator=g)  # bounds stay the unit box; everything else in ONE cell
    centres = torch.rand(8, 3, generator=g)
    clustered = centres[torch.randint(0, 8, (P,), generator=g)] + 1e-4 * torch.randn(P, 3, generator=g)
    for pts in (same.cuda(), clustered.cuda()):
        a = _lib.morton_order(pts, min_points=0)
        b = _lib.morton_order(pts, min_points=0)
        assert torch.equal(a, b)
        o = a.cpu().numpy()
        assert np.array_equal(np.sort(o), np.arange(P))
        cells = curve_cells(pts.cpu().numpy(), 21 if P >= (1 << 20) else 18)[o]
        assert (np.diff(cells) >= 0).all()
        assert (np.diff(o)[np.diff(cells) == 0] > 0).all()
