"""MI355X-native batched SDF query engine with the pytorch_volumetric API (drop-in for the SDF-query hot path).

Every name pytorch_volumetric exports for that path is available here; see INTEGRATION.md for the differences.
"""
from pytorch_volumetric_amd import _lib, mesh_io, voxel
from pytorch_volumetric_amd._lib import RULE_RES_F64, RULE_ROUND_FLOOR_HALF, RULE_ROUND_HALF_AWAY, RULE_VALID_ON_INDEX
from pytorch_volumetric_amd.sdf import (CachedSDF, ComposedSDF, MeshObjectFactory, MeshSDF, ObjectFactory,
                                        ObjectFrameSDF, OutOfBoundsStrategy, PreparedPoints, SDFQuery, SphereSDF,
                                        sample_mesh_points)
from pytorch_volumetric_amd.model_to_sdf import RobotSDF, aabb_to_ordered_end_points, cache_link_sdf_factory
from pytorch_volumetric_amd.chamfer import (PlausibleDiversity, batch_chamfer_dist, pairwise_distance,
                                            pairwise_distance_chamfer)
from pytorch_volumetric_amd.voxel import get_coordinates_and_points_in_grid, get_divisible_range_by_resolution
from pytorch_volumetric_amd.voxel_containers import (ExpandingVoxelGrid, VoxelGrid, VoxelSet, Voxels,
                                                     voxel_down_sample)
from pytorch_volumetric_amd.volume import is_inside
from pytorch_volumetric_amd.visualization import draw_sdf_slice, get_transformed_meshes
# stand-ins for the pytorch_kinematics members the path touches, and the multi-GPU helpers
from pytorch_volumetric_amd.transforms import Rotate, Transform3d, Translate
from pytorch_volumetric_amd.kinematics import Chain, build_chain_from_urdf, build_serial_chain_from_urdf
from pytorch_volumetric_amd.dist import ShardedSDF, shard_range, sharded_chamfer
from pytorch_volumetric_amd.warmup import warm_up

__version__ = "0.1.0"
