"""-m gpu: the library keeps no mutable state and the Python objects are shared safely -- several host threads, each on its own
HIP stream, query the SAME CachedSDF / ComposedSDF / MeshSDF objects at once (ctypes releases the GIL inside every entry point,
so the launches really interleave) and every result equals the single-threaded one, bit for bit (SURVEY.md section 8(b):
"re-entrant from multiple Python threads / one stream per GPU")."""
import threading

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_four_threads_on_four_streams_share_the_sdf_objects():
    gt = H.drill_like_gt()
    leaves = [pv.CachedSDF("leaf", 0.01, H.padded_range(H.DRILL_BB, 0.1, as_numpy=(s % 2 == 0)), gt, device="cuda", cache_path=None)
              for s in range(4)]
    A = 3
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=H.random_rigid(4 * A, seed=5, trans=0.3)), batch_dim=(A,))
    comp.group_points = True
    mesh = pv.MeshSDF(pv.MeshObjectFactory(H.mesh_path("probe.obj")))
    T, ROUNDS = 4, 12
    chunk = _lib.group_chunk_points()
    sizes = [chunk + 17, 2 * chunk + 1, 5000, 777]
    inputs = [[H.uniform_points(sizes[(k + r) % 4], [-0.5] * 3, [0.5] * 3, seed=100 * k + r).cuda() for r in range(3)] for k in range(T)]
    mesh_pts = [H.uniform_points(3000 + 100 * k, [-0.1] * 3, [0.1] * 3, seed=900 + k).cuda() for k in range(T)]

    def one_round(k, r):
        p = inputs[k][r % 3]
        v, g = leaves[k % 4](p)
        cv, cg = comp(p)
        P = p.shape[0]
        qv, qg = torch.empty((A, P), device="cuda"), torch.empty((A, P, 3), device="cuda")
        comp.query_into(p, qv, qg)
        mv, mg = mesh(mesh_pts[k])
        return [t.clone() for t in (v, g, cv, cg, qv, qg, mv, mg)]

    want = [[one_round(k, r) for r in range(3)] for k in range(T)]
    torch.cuda.synchronize()
    got = [[None] * ROUNDS for _ in range(T)]
    errors = []
    gate = threading.Barrier(T)

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            gate.wait()
            with torch.cuda.stream(stream):
                for r in range(ROUNDS):
                    got[k][r] = one_round(k, r)
            stream.synchronize()
        except Exception as exc:  # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors and not any(t.is_alive() for t in threads), errors
    torch.cuda.synchronize()
    for k in range(T):
        for r in range(ROUNDS):
            for a, b in zip(got[k][r], want[k][r % 3]):
                assert np.array_equal(a.cpu().numpy().view(np.int32), b.cpu().numpy().view(np.int32)), (k, r)
