"""VoxelGrid scatter/gather and voxel_down_sample on the HIP kernels vs the generic torch path on the same GPU."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import voxel_containers as vc


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


g = torch.Generator(device="cuda").manual_seed(0)
P = 10_000_000
cloud = torch.randn((P, 3), generator=g, device="cuda") * torch.tensor([0.3, 0.2, 0.1], device="cuda")
vals = torch.randn((P,), generator=g, device="cuda")
grid = pv.VoxelGrid(0.01, [(-1.5, 1.5), (-1.0, 1.0), (-0.5, 0.5)], device="cuda")
t_set = timed(lambda: grid.__setitem__(cloud, vals))
t_get = timed(lambda: grid[cloud])
t_ds = timed(lambda: pv.voxel_down_sample(cloud, 0.01))
orig = vc.ValueRangeView._device_path
vc.ValueRangeView._device_path = lambda self, pts: False  # generic torch path, same GPU
t_set_t = timed(lambda: grid.__setitem__(cloud, vals))
t_get_t = timed(lambda: grid[cloud])
t_ds_t = timed(lambda: pv.voxel_down_sample(cloud, 0.01))
vc.ValueRangeView._device_path = orig
nvox = grid.get_voxel_values().numel()
print(f"{P} points, grid {tuple(grid.get_voxel_values().shape)} = {nvox} voxels")
print(f"scatter (per-point values): HIP {t_set*1e3:.2f} ms ({P/t_set:.2e} pts/s, {16*P/t_set/1e9:.0f} GB/s of 16 B/pt)   torch {t_set_t*1e3:.2f} ms")
print(f"gather:                     HIP {t_get*1e3:.2f} ms ({P/t_get:.2e} pts/s, {16*P/t_get/1e9:.0f} GB/s of 16 B/pt)   torch {t_get_t*1e3:.2f} ms")
print(f"voxel_down_sample(0.01):    HIP {t_ds*1e3:.2f} ms   torch {t_ds_t*1e3:.2f} ms")
