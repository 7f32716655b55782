"""One voxel-cache build (drill, 0.002 m, 5 cm padding: 2.16 M centres x 15,728 triangles) for rocprofv3 passes."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
drill = pv.MeshObjectFactory(os.path.join("tests", "golden", "meshes", "ycb_power_drill.npz"))
res = float(sys.argv[1]) if len(sys.argv) > 1 else 0.002
for _ in range(2):
    c = pv.CachedSDF("drill", res, drill.bounding_box(padding=0.05), pv.MeshSDF(drill), clean_cache=True,
                     cache_path="/tmp/run_build_cache.pkl")
torch.cuda.synchronize()
print(c.voxels.shape)
