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


def test_composed_sdf_reproduces_the_reference_call():
    c = cached_from_golden("f64")
    pts = torch.from_numpy(G["composed/points"]).cuda().reshape(3, 500, 3)
    comp = pv.ComposedSDF([c, c, c], torch.from_numpy(G["composed/single/tf"]))
    v1, g1 = comp(pts)
    assert v1.shape == (1500,) and g1.shape == (1500, 3)  # flat, like the reference without a transform batch
    comp.set_transforms(torch.from_numpy(G["composed/batched/tf"]), batch_dim=(4,))
    v2, g2 = comp(pts)
    assert v2.shape == (4, 3, 500) and g2.shape == (4, 3, 500, 3)
    for v, g, name in ((v1, g1, "single"), (v2, g2, "batched")):
        rv, rg = G[f"composed/{name}/val"], G[f"composed/{name}/grad"]
        close = np.isclose(v.cpu().numpy(), rv, rtol=0, atol=1e-6)
        assert close.mean() > 0.995
        gc = np.isclose(g.cpu().numpy()[close], rg[close], rtol=0, atol=2e-6) | np.isnan(rg[close])
        assert gc.mean() > 0.999
    assert np.abs(v2.cpu().numpy() - G["composed/batched/val"]).max() < 0.06  # a boundary crossing moves one voxel at most
