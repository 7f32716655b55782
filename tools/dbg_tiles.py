import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import mesh_io, _lib
from tests import helpers as H
m = mesh_io.uv_sphere_mesh(0.1, 250, 200)
obj = pv.MeshObjectFactory(mesh=m)
obj._mesh_desc()
tiles = obj._tiles_dev.cpu().numpy()[: (m.faces.shape[0] + 255) // 256]
rec = obj._rec_dev.cpu().numpy()
print("ntiles", len(tiles), "tile r: min/mean/max", tiles[:, 3].min(), tiles[:, 3].mean(), tiles[:, 3].max())
print("tri r: min/mean/max", rec[:, 3].min(), rec[:, 3].mean(), rec[:, 3].max())
pts = H.uniform_points(1 << 16, [-0.15] * 3, [0.15] * 3, seed=2)
order = _lib.morton_order(pts.cuda()).cpu().long()
sp = pts[order].numpy()
mtrue = np.abs(np.linalg.norm(sp, axis=1) - 0.1)
d = np.linalg.norm(sp[:, None, :] - tiles[None, :, :3], axis=2)
need = (d - tiles[None, :, 3]) <= mtrue[:, None] * 1.0001 + 1e-4
print("needed tiles per point: mean", need.sum(1).mean(), "median", np.median(need.sum(1)))
wave_need = need.reshape(-1, 64, need.shape[1]).any(axis=1)
print("needed tiles per wave (64 sorted points): mean", wave_need.sum(1).mean(), "of", need.shape[1])
spread = np.linalg.norm(sp.reshape(-1, 64, 3).max(1) - sp.reshape(-1, 64, 3).min(1), axis=1)
print("wave bbox diagonal mean", spread.mean())
