"""The optional host-call module (csrc/fastcall.cpp -> pytorch_volumetric_amd/_pvamd_fast.so): host plumbing for the drop-in fast
paths -- the same checks, the same two allocations and the same C-ABI entry point of the same libpvamd.so, without the interpreter
in between.  With it and without it (the ctypes path) every call returns the same tensors, bit for bit; whatever is not its case
(other dtypes, strides, devices, shapes) is handed back to the Python path, which converts or raises as before."""
import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from tests import helpers as H


def test_the_module_is_optional_and_names_its_three_calls():
    fast = _lib.fastcall()
    if fast is None:
        pytest.skip("pytorch_volumetric_amd/_pvamd_fast.so not built (make -C pytorch_volumetric_amd/csrc fast)")
    assert {"cached_call", "cached_into", "composed_call"} <= set(dir(fast))
    # not its case (a CPU tensor): handed back, nothing launched
    assert fast.cached_call(0, 0, 0, torch.zeros((4, 3))) is None
    assert fast.cached_into(0, 0, 0, torch.zeros((4, 3)), torch.zeros((4,)), torch.zeros((4, 3))) is False
    assert fast.composed_call(0, 0, 1, 0, 1, (), 0, 0, torch.zeros((4, 3))) is None


class _without_fastcall:
    """the ctypes path: plans are rebuilt without the module, and again with it afterwards"""

    def __enter__(self):
        self.was = _lib._fast
        _lib._fast = None
        _lib.EPOCH[0] += 1

    def __exit__(self, *exc):
        _lib._fast = self.was
        _lib.EPOCH[0] += 1


def _same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and a.device == b.device and a.is_contiguous() == b.is_contiguous() and \
        np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)


@pytest.mark.gpu
def test_with_and_without_the_module_every_call_returns_the_same_tensors():
    fast = _lib.fastcall()
    if fast is None:
        pytest.skip("pytorch_volumetric_amd/_pvamd_fast.so not built")
    gt = H.drill_like_gt()
    cached = pv.CachedSDF("leaf", 0.01, H.padded_range(H.DRILL_BB, 0.1), gt, device="cuda", cache_path=None)
    lo, hi = [r[0] - 0.05 for r in cached.ranges], [r[1] + 0.05 for r in cached.ranges]
    base = H.uniform_points(40_000, lo, hi, seed=11).cuda()
    base[17] = float("nan")
    base[18, 1] = float("inf")
    inputs = {
        "flat": base,
        "empty": base[:0],
        "one": base[:1],
        "leading dims": base[:39_000].reshape(13, 3000, 3),
        "strided (Python path)": base[::2],
        "float64 (Python path)": base[:500].double(),
        "cpu (Python path)": base[:500].cpu(),
    }
    for name, p in inputs.items():
        v1, g1 = cached(p)
        with _without_fastcall():
            v2, g2 = cached(p)
        assert _same(v1, v2) and _same(g1, g2), name
    # query_into, incl. the calls it must refuse exactly as before
    P = 33_333
    p = base[:P].contiguous()
    v1, g1 = torch.empty((P,), device="cuda"), torch.empty((P, 3), device="cuda")
    v2, g2 = torch.empty_like(v1), torch.empty_like(g1)
    cached.query_into(p, v1, g1)
    with _without_fastcall():
        cached.query_into(p, v2, g2)
    assert _same(v1, v2) and _same(g1, g2)
    for bad in (lambda: cached.query_into(p, v1[:-1], g1), lambda: cached.query_into(p, v1, g1[:, :2]),
                lambda: cached.query_into(p.double(), v1, g1), lambda: cached.query_into(base[::2][:P], v1, g1)):
        with pytest.raises((ValueError, _lib.PvamdError)):
            bad()
    # a reassigned box is seen by the module's calls too (the descriptor address travels in the plan)
    cached.bb = cached.bb + 0.01
    v3, g3 = cached(p)
    cached.query_into(p, v1, g1)
    with _without_fastcall():
        v4, g4 = cached(p)
    assert _same(v3, v4) and _same(g3, g4) and _same(v1, v4) and _same(g1, g4) and not _same(v3, v2)

    # compositions: no batch (flat results), one and two batch dimensions, leading point dimensions
    leaves = [pv.CachedSDF(f"leaf{s}", 0.01, H.padded_range(H.DRILL_BB, 0.1), gt, device="cuda", cache_path=None) for s in range(3)]
    pts = H.uniform_points(6000, [-0.5] * 3, [0.5] * 3, seed=5).cuda()
    for batch in (None, (4,), (2, 3)):
        A = 1 if batch is None else int(np.prod(batch))
        comp = pv.ComposedSDF(leaves, None)
        comp.set_transforms(pv.Transform3d(matrix=H.random_rigid(3 * A, seed=7 + A, trans=0.3)), batch_dim=batch)
        for q in (pts, pts.reshape(3, 2000, 3), pts[:1]):
            c1 = comp(q)
            with _without_fastcall():
                c2 = comp(q)
            assert _same(c1[0], c2[0]) and _same(c1[1], c2[1]), (batch, tuple(q.shape))
            expect = (q.numel() // 3,) if batch is None else (*batch, *q.shape[:-1])
            assert tuple(c1[0].shape) == expect and tuple(c1[1].shape) == (*expect, 3)


@pytest.mark.gpu
def test_an_error_code_of_the_entry_point_is_raised():
    fast = _lib.fastcall()
    if fast is None:
        pytest.skip("pytorch_volumetric_amd/_pvamd_fast.so not built")
    import ctypes
    desc = _lib.GridDesc()  # zeroed: pvamd_cached_query refuses it (no launch)
    p = torch.zeros((64, 3), device="cuda")
    with pytest.raises(_lib.PvamdError, match="pvamd_cached_query: invalid argument"):
        fast.cached_call(_lib.entry_address("pvamd_cached_query"), ctypes.addressof(desc), torch.cuda.current_device(), p)
