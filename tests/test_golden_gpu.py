"""-m gpu: the HIP path against the committed golden vectors produced by the reference's own code
(tests/golden/make_golden.py): CachedSDF / ComposedSDF glue run verbatim over shims of the absent third-party
packages.  Same tolerances as tests/test_oracle_pinned.py (which holds the oracle to the same vectors)."""
import os

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(H.GOLDEN, "reference_lifted.npz"))


class StoredGT(pv.ObjectFrameSDF):
    """gt_sdf that replays the reference's stored grid (so the CachedSDF under test holds exactly its values)."""

    def __init__(self, tag):
        self.val, self.grad, self.bb = G[f"cached/{tag}/val_grid"], G[f"cached/{tag}/grad_grid"], G[f"cached/{tag}/bb"]

    def __call__(self, pts):
        return torch.from_numpy(self.val).reshape(-1), torch.from_numpy(self.grad)

    def surface_bounding_box(self, **kw):
        return torch.from_numpy(self.bb)


def cached_from_golden(tag):
    rng = G[f"cached/{tag}/range_snapped"]
    rng = rng if tag == "f64" else [(float(a), float(b)) for a, b in rng]
    c = pv.CachedSDF("sphere", 0.05, rng, StoredGT(tag), device="cuda", cache_path=None)
    assert np.array_equal(np.array(c.ranges, dtype=np.float64), G[f"cached/{tag}/range_snapped"])
    return c


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_cached_sdf_reproduces_the_reference_call(tag):
    c = cached_from_golden(tag)
    assert c._view.index_f64 == (tag == "f64")
    pts = torch.from_numpy(G[f"cached/{tag}/points"]).cuda()
    assert np.array_equal(c.voxels.ensure_index_key(pts).cpu().numpy(), G[f"cached/{tag}/keys"])  # bit-exact indices
    valid = c.voxels.get_valid_values(pts).cpu().numpy()
    assert np.array_equal(valid, G[f"cached/{tag}/valid"])
    val, grad = c(pts)
    val, grad = val.cpu().numpy(), grad.cpu().numpy()
    rv, rg = G[f"cached/{tag}/val"], G[f"cached/{tag}/grad"]
    assert np.array_equal(val[valid], rv[valid]) and np.array_equal(grad[valid], rg[valid])
    oob = ~valid
    assert np.abs(val[oob] - rv[oob]).max() <= 1.2e-7
    fin = oob & np.isfinite(rg).all(axis=1)
    assert np.abs(grad[fin] - rg[fin]).max() <= 2.4e-7
    assert np.array_equal(np.isnan(grad), np.isnan(rg))
    assert np.array_equal(c.outside_surface(pts, 0.02).cpu().numpy(), G[f"cached/{tag}/outside"])
    vb, gb = c(pts[:600].reshape(2, 3, 100, 3))
    assert list(vb.shape) == list(G[f"cached/{tag}/batched_shape"]) and gb.shape == (2, 3, 100, 3)


def leaf_value_ties(c, tf, A, pts, tol=2e-6):
    """(A, P) mask: the two smallest per-leaf values of a 3-leaf composition of `c` are within tol of each other."""
    tf = torch.as_tensor(np.asarray(tf), dtype=torch.float32).reshape(3, A, 4, 4).cuda()
    p = torch.as_tensor(np.asarray(pts), dtype=torch.float32).reshape(-1, 3).cuda()
    vals = []
    for s in range(3):
        x = p.unsqueeze(0) @ tf[s, :, :3, :3].transpose(-1, -2) + tf[s, :, None, :3, 3]
        vals.append(c(x)[0])
    two = torch.stack(vals).sort(dim=0).values[:2]
    return ((two[1] - two[0]).abs() <= tol).cpu().numpy()


def test_composed_sdf_reproduces_the_reference_call():
    c = cached_from_golden("f64")
    pts = torch.from_numpy(G["composed/points"]).cuda().reshape(3, 500, 3)
    comp = pv.ComposedSDF([c, c, c], torch.from_numpy(G["composed/single/tf"]))
    v1, g1 = comp(pts)
    assert v1.shape == (1500,) and g1.shape == (1500, 3)  # flat, like the reference without a transform batch
    comp.set_transforms(torch.from_numpy(G["composed/batched/tf"]), batch_dim=(4,))
    v2, g2 = comp(pts)
    assert v2.shape == (4, 3, 500) and g2.shape == (4, 3, 500, 3)
    for v, g, name, A in ((v1, g1, "single", 1), (v2, g2, "batched", 4)):
        rv, rg = G[f"composed/{name}/val"], G[f"composed/{name}/grad"]
        # 1e-6 everywhere, except where the shims' torch matmul and the kernel's fma chain -- a last-place difference in a
        # leaf-frame coordinate -- land on different sides of a half-voxel plane or a range edge; every such point is
        # accounted for individually (no percentage)
        rep = {}
        n_bad, n_unexplained = H.composed_disagreements_explained([c, c, c], G[f"composed/{name}/tf"], A,
                                                                  G["composed/points"], v.cpu().numpy(), rv, report=rep)
        print(f"composed/{name}: {n_bad} of {rv.size} values differ by more than 1e-6, all within {rep['max_units_needed']:.2f} "
              "rounding units (bound: 8) of a voxel / range boundary")
        assert n_unexplained == 0 and n_bad < 0.01 * rv.size
        close = np.isclose(v.cpu().numpy(), rv, rtol=0, atol=1e-6)
        # gradients of agreeing values: 2e-6, except where two leaves tie in value (either leaf's gradient is a valid
        # first-minimum under 1-ulp value differences); ties are detected from the per-leaf values
        gv = g.cpu().numpy()
        gclose = np.isclose(gv, rg, rtol=0, atol=2e-6).all(axis=-1) | np.isnan(rg).any(axis=-1)
        tie = leaf_value_ties(c, G[f"composed/{name}/tf"], A, G["composed/points"]).reshape(close.shape)
        assert not (close & ~gclose & ~tie).any()
    assert np.abs(v2.cpu().numpy() - G["composed/batched/val"]).max() < 0.06  # a boundary crossing moves one voxel at most
