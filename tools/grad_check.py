"""A/B of the out-of-range gradient statements (PVAMD_GRAD_EXPERIMENT builds): max ulp distance to the product library's bits."""
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
import workloads as Wk
cached = Wk.build_c2_cache()
pts = Wk.c2_points(cached, 1 << 22, seed=5, margin=0.5)
v, g = cached(pts)
out = os.environ.get("GRAD_REF")
if out and not os.path.exists(out):
    np.savez(out, v=v.cpu().numpy(), g=g.cpu().numpy())
    print("reference written")
else:
    z = np.load(out)
    gv, gg = v.cpu().numpy(), g.cpu().numpy()
    same_v = np.array_equal(gv.view(np.int32), z["v"].view(np.int32))
    d = np.abs(gg.view(np.int32).astype(np.int64) - z["g"].view(np.int32).astype(np.int64))
    nan_same = np.array_equal(np.isnan(gg), np.isnan(z["g"]))
    d = np.where(np.isnan(gg), 0, d)
    print(os.environ.get("PVAMD_LIB", "product"), "values same bits", same_v, "| NaN pattern same", nan_same, "| grad max ulp", int(d.max()), "| differing", int((d > 0).sum()), "of", d.size)
