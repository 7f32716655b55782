"""-m gpu: the reference README's code blocks, line for line with `pytorch_volumetric_amd as pv` (README.md:31-200 of the
reference) -- the drill as the committed .npz instead of the YCB .obj, `pv.Translate` / `pv.build_serial_chain_from_urdf` where the
README imports pytorch_kinematics, a synthetic 7-joint arm where it loads the KUKA (its assets are not available offline)."""
import math
import os

import numpy as np
import pytest
import torch

import pytorch_volumetric_amd as pv
from tests import helpers as H
from tests.test_robot_gpu import synthetic_arm

pytestmark = pytest.mark.gpu


def test_readme_blocks_run_as_written(tmp_path):
    # "SDF from mesh" (README.md:31-38)
    obj = pv.MeshObjectFactory(H.mesh_path("ycb_power_drill.npz"))
    sdf = pv.MeshSDF(obj)
    # "Cached SDF" (README.md:44-48): same call, the cache file in a scratch directory
    cached_sdf = pv.CachedSDF('drill', resolution=0.01, range_per_dim=obj.bounding_box(padding=0.1), gt_sdf=sdf,
                              cache_path=str(tmp_path / "sdf_cache.pkl"))
    assert os.path.exists(tmp_path / "sdf_cache.pkl")
    # out_of_bounds_strategy=pv.OutOfBoundsStrategy.LOOKUP_GT_SDF (README.md:54)
    exact_outside = pv.CachedSDF('drill', resolution=0.01, range_per_dim=obj.bounding_box(padding=0.1), gt_sdf=sdf,
                                 out_of_bounds_strategy=pv.OutOfBoundsStrategy.LOOKUP_GT_SDF, cache_path=str(tmp_path / "sdf_cache.pkl"))
    # "Composed SDF" (README.md:60-70)
    sdf1, sdf2 = pv.MeshSDF(obj), pv.MeshSDF(obj)
    tsf1, tsf2 = pv.Translate(0.1, 0, 0), pv.Translate(-0.2, 0, 0.2)
    composed = pv.ComposedSDF([sdf1, sdf2], tsf1.stack(tsf2))
    # "SDF value and gradient queries" (README.md:78-92)
    query_range = np.array([[-1, 0.5], [-0.5, 0.5], [-0.2, 0.8]])
    coords, pts = pv.get_coordinates_and_points_in_grid(0.05, query_range)  # (0.01 in the README: 1.5 M points; same code path)
    sdf_val, sdf_grad = composed(pts)
    assert sdf_val.shape == (len(pts),) and sdf_grad.shape == (len(pts), 3) and sdf_val.device.type == "cpu"
    # the transforms map the object frame INTO each leaf's frame (obj_frame_to_each_frame, sdf.py:333-345)
    v1, _ = sdf(pts + torch.tensor([0.1, 0.0, 0.0]))
    v2, _ = sdf(pts + torch.tensor([-0.2, 0.0, 0.2]))
    assert torch.allclose(sdf_val, torch.minimum(v1, v2), atol=1e-6)
    far = H.uniform_points(2000, [0.5, 0.5, 0.5], [0.9, 0.9, 0.9], seed=1)
    assert torch.allclose(exact_outside(far)[0], sdf(far)[0], atol=1e-6)  # outside the cache: the ground truth
    assert (cached_sdf(far)[0] <= sdf(far)[0] + 1e-5).all()               # ... or the distance to the bounding box (a lower bound)
    # "Plotting a 2D slice" (README.md:100-115), headless
    import matplotlib
    matplotlib.use("Agg")
    ret = pv.draw_sdf_slice(sdf, np.array([[-0.15, 0.2], [0, 0], [-0.1, 0.2]]))
    assert ret[6].shape == (31, 36) and ret[3] is not None
    # "Robot SDF" (README.md:125-200)
    chain = synthetic_arm(str(tmp_path))
    d = "cuda" if torch.cuda.is_available() else "cpu"
    chain = chain.to(device=d)
    s = pv.RobotSDF(chain, path_prefix=str(tmp_path))
    s = pv.RobotSDF(chain, path_prefix=str(tmp_path),
                    link_sdf_cls=pv.cache_link_sdf_factory(resolution=0.02, padding=1.0, device=d, cache_path=None))
    th = torch.tensor([0.0, -math.pi / 4.0, 0.0, math.pi / 2.0, 0.0, math.pi / 4.0, 0.0], device=d)
    N = 200
    th_perturbation = torch.randn(N - 1, 7, device=d) * 0.1
    th = torch.cat((th.view(1, -1), th_perturbation + th))
    y = 0.02
    query_range = np.array([[-1, 0.5], [y, y], [-0.2, 0.8]])
    coords, pts = pv.get_coordinates_and_points_in_grid(0.01, query_range, device=s.device)
    s.set_joint_configuration(th)
    sdf_val, sdf_grad = s(pts)
    assert pts.shape == (15251, 3)  # the M of the README's timing table
    assert sdf_val.shape == (N, 15251) and sdf_grad.shape == (N, 15251, 3) and sdf_val.device.type == "cuda"
    s.set_joint_configuration(th[0])
    single, _ = s(pts)
    assert single.shape == (15251,) and torch.equal(single, sdf_val[0])
    meshes = pv.get_transformed_meshes(s)
    assert len(meshes) == 8
