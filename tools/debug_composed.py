"""Where does the new wave-tile leaf loop differ from the oracle?  Mismatch census on two failing test scenes."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
from oracle import oracle
from tests import helpers as H
from tests.test_composed_queue_gpu import make_leaf, query_with_leaf_ids


def census(name, leaves, tfm, A, pts):
    comp = pv.ComposedSDF(leaves, None)
    comp.set_transforms(pv.Transform3d(matrix=tfm), batch_dim=(A,) if A > 1 else None)
    ogr = [H.oracle_grid_from_cached(l) for l in leaves]
    oval, ograd, oleaf = oracle.composed_query(ogr, tfm.numpy(), A, pts.numpy())
    S = len(leaves)
    # per-leaf in-range flags of configuration 0 from the oracle pieces
    m = tfm.reshape(S, A, 4, 4).numpy()
    for flags in (4, 20):
        val, grad, leaf = query_with_leaf_ids(comp, pts, flags)
        bad = ~((val == oval) | (np.isnan(val) & np.isnan(oval)))
        print(f"{name} flags {flags}: {bad.sum()} of {bad.size} values differ; leaf ids differ {(leaf != oleaf).sum()}")
        if bad.sum():
            a, i = np.argwhere(bad)[0]
            idx = np.argwhere(bad)
            print("   first bad (a, point):", a, i, "gpu", val[a, i], leaf[a, i], "oracle", oval[a, i], oleaf[a, i])
            print("   bad per configuration:", np.bincount(idx[:, 0], minlength=A)[:8], " bad point positions mod 256 (hist of 8 bins):",
                  np.histogram(idx[:, 1] % 256, bins=8, range=(0, 256))[0], " tiles touched:", len(np.unique(idx[:, 1] // 256)))
            gl, ol = leaf[bad], oleaf[bad]
            print("   gpu > oracle (missed a candidate):", (val[bad] > oval[bad]).sum(), " gpu < oracle (invented one):", (val[bad] < oval[bad]).sum())
            # was the oracle's winner an in-range lookup?  (query that leaf alone)
            for (a, i) in idx[:6]:
                s = oleaf[a, i]
                x = m[s, a, :3, :3] @ pts[i].numpy().astype(np.float32) + m[s, a, :3, 3]
                v1, g1, oob = oracle.cached_query(ogr[s], x[None].astype(np.float32))
                s2 = leaf[a, i]
                x2 = m[s2, a, :3, :3] @ pts[i].numpy().astype(np.float32) + m[s2, a, :3, 3]
                v2, _, oob2 = oracle.cached_query(ogr[s2], x2[None].astype(np.float32))
                print(f"      a={a} i={i} (lane {i % 64}, k {(i % 256) // 64}): oracle leaf {s} val {oval[a, i]:.6g} oob={bool(oob[0])} | gpu leaf {s2} val {val[a, i]:.6g}; that leaf alone gives {v2[0]:.6g} oob={bool(oob2[0])}")


# scene 1: S = 70 sparse
leaf = make_leaf(res=0.02)
S, A = 70, 16
tfm = H.random_rigid(S * A, seed=9, trans=0.8)
census("S=70", [leaf] * S, tfm, A, H.uniform_points(65_536 + 4, [-1.0] * 3, [1.0] * 3, seed=4))
# scene 2: 4 overlapping leaves, random order
S, A = 4, 3
leaves = [make_leaf(f64=(s % 2 == 0), res=0.02, padding=0.03) for s in range(S)]
tfm = H.random_rigid(S * A, seed=2, trans=0.1)
census("S=4 overlap", leaves, tfm, A, H.uniform_points(8192, [-0.25] * 3, [0.25] * 3, seed=5))
