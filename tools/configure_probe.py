"""Where do the ~15 us of robot.set_joint_configuration(q_gpu) go on the host?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib, transforms as tf
import workloads as Wk
robot = Wk.build_c4(0.02, 0.1)
for A in (20, 200):
    q = Wk.c4_joint_configs(A).cuda().float().contiguous()
    robot.set_joint_configuration(q); torch.cuda.synchronize()
    pv.warm_up()
    def wall(fn, n=20000):
        for _ in range(200): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    lib = _lib.load(); dev = _lib.require_gpu(); S = len(robot.sdf_to_link_name); M = len(robot.joint_names)
    off = robot._offset_inv_dev(dev)
    stack = robot._configure(lib, dev, q, A, M, S, off)
    print(f"A={A}: whole call {wall(lambda: robot.set_joint_configuration(q)):.2f} us | _configure alone (allocates the stack) "
          f"{wall(lambda: robot._configure(lib, dev, q, A, M, S, off)):.2f} | _configure into a given stack "
          f"{wall(lambda: robot._configure(lib, dev, q, A, M, S, off, stack=stack)):.2f} | Transform3d(matrix=stack) "
          f"{wall(lambda: tf.Transform3d(matrix=stack)):.2f} | sdf.set_transforms "
          f"{wall(lambda: robot.sdf.set_transforms(robot.object_to_link_frames, batch_dim=robot.configuration_batch, known_rigid=True)):.2f}", flush=True)
