"""MI355X-native batched SDF query engine with the pytorch_volumetric API (drop-in for the SDF-query hot path).

Exports the reference's names (pytorch_volumetric/__init__.py:1-9) that lie on that path; the voxel containers,
visualisation helpers and point-cloud down-sampling of the reference are out of scope.
"""
from pytorch_volumetric_amd.chamfer import batch_chamfer_dist, PlausibleDiversity, pairwise_distance, \
    pairwise_distance_chamfer
from pytorch_volumetric_amd.sdf import sample_mesh_points, ObjectFrameSDF, MeshSDF, CachedSDF, ComposedSDF, SDFQuery, \
    ObjectFactory, MeshObjectFactory, OutOfBoundsStrategy, SphereSDF
from pytorch_volumetric_amd.voxel import Voxels, VoxelGrid, VoxelSet, ExpandingVoxelGrid, get_divisible_range_by_resolution, \
    get_coordinates_and_points_in_grid, voxel_down_sample
from pytorch_volumetric_amd.model_to_sdf import RobotSDF, cache_link_sdf_factory, aabb_to_ordered_end_points
from pytorch_volumetric_amd.volume import is_inside
from pytorch_volumetric_amd.transforms import Transform3d, Translate, Rotate
from pytorch_volumetric_amd.kinematics import Chain, build_chain_from_urdf, build_serial_chain_from_urdf
from pytorch_volumetric_amd.dist import ShardedSDF, sharded_chamfer, shard_range

__version__ = "0.1.0"
