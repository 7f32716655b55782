"""Wall times of the set-up flows around the hot path (one MI355X): RobotSDF construction with cached links, surface sampling,
bounding boxes."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import torch, pytorch_volumetric_amd as pv, workloads as Wk
from tests import helpers as H


def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for pad in (0.1, 1.0):
    print(f"RobotSDF(chain, cache_link_sdf_factory(0.02, padding={pad})): %.1f ms (8 links, {'1.3 M' if pad == 1.0 else '6 k'} voxels each)" % wall(lambda: Wk.build_c4(0.02, pad)))
robot = Wk.build_c4(0.02, 0.1)
robot.set_joint_configuration(Wk.c4_joint_configs(200))
print("robot.surface_bounding_box() A=200: %.2f ms | link_bounding_boxes(): %.2f ms" % (wall(lambda: robot.surface_bounding_box()), wall(lambda: robot.link_bounding_boxes())))
obj = Wk.build_drill()
for n in (500, 1000, 100000):
    print(f"sample_mesh_points(drill, num_points={n}): %.2f ms" % wall(lambda: pv.sample_mesh_points(obj, num_points=n, name="d", dbpath=None)))
print("MeshObjectFactory(drill) + first query (upload + prepare): %.2f ms" % wall(lambda: pv.MeshSDF(pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz")))(torch.zeros(64, 3, device="cuda"))))
