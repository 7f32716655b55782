"""Callers of the query path that draw (visualization.py:21-107 of the reference): a planar slice of an SDF, and the link
meshes of a RobotSDF placed by its current configuration.  Thin on purpose -- the slice is ONE batched `__call__` over the
grid the reference builds (voxel.get_coordinates_and_points_in_grid), matplotlib is imported only when something is drawn,
and the meshes come back as `mesh_io.TriMesh` (this package's stand-in for open3d's TriangleMesh)."""
import numpy as np
import torch

from pytorch_volumetric_amd import transforms as tf
from pytorch_volumetric_amd import voxel


def _level_label(x):
    """label of a contour level: the zero level is the surface"""
    if x == 0:
        return "surface"
    s = f"{x:.1f}"
    return f"{x:.0f}" if s.endswith("0") else s


def draw_sdf_slice(s, query_range, resolution=0.01, interior_padding=0.2, cmap="Greys_r", device="cpu", plot_grad=False,
                   do_plot=True):
    """visualization.py:21-80.  `query_range`: (min, max) per dimension in the SDF's frame, exactly one dimension with
    min == max (the slicing plane).  Returns (sdf_val, sdf_grad, pts, ax, cset1, cset2, v) like the reference; with
    do_plot=False nothing is imported or drawn and ax / cset1 / cset2 are None."""
    coords, pts = voxel.get_coordinates_and_points_in_grid(resolution, query_range, device=device)
    # the reference jitters the regular grid by 1e-6 (global torch RNG) so that no point sits on a voxel border
    pts = pts + torch.randn_like(pts) * 1e-6
    dim_labels = ['x', 'y', 'z']
    slice_dim = next((i for i in range(len(dim_labels)) if len(coords[i]) == 1), None)
    if slice_dim is None:
        raise RuntimeError("Sliced SDF requires a single query value for the sliced, but all query dimensions > 1")
    shown = [i for i in range(3) if i != slice_dim]

    sdf_val, sdf_grad = s(pts)
    x, z = coords[shown[0]].cpu(), coords[shown[1]].cpu()
    v = sdf_val.reshape(len(x), len(z)).transpose(0, 1).cpu()
    ax = cset1 = cset2 = None
    if do_plot:
        import matplotlib.colors
        import matplotlib.pyplot as plt
        norm = matplotlib.colors.Normalize(vmin=float(sdf_val.min().cpu()) - interior_padding, vmax=float(sdf_val.max().cpu()))
        ax = plt.gca()
        ax.set_xlabel(dim_labels[shown[0]])
        ax.set_ylabel(dim_labels[shown[1]])
        cset1 = ax.contourf(x, z, v, norm=norm, cmap=cmap)
        cset2 = ax.contour(x, z, v, colors='k', levels=[0], linestyles='dashed')
        if plot_grad:
            uv = sdf_grad.reshape(len(x), len(z), 3).permute(1, 0, 2).cpu()
            n = 5  # every fifth arrow
            ax.quiver(x[::n], z[::n], uv[::n, ::n, shown[0]], uv[::n, ::n, shown[1]], color='g')
        ax.clabel(cset2, cset2.levels, inline=True, fontsize=13, fmt=_level_label)
        plt.colorbar(cset1)
        plt.draw()
        plt.pause(0.005)
    return sdf_val, sdf_grad, pts, ax, cset1, cset2, v


def get_transformed_meshes(robot_sdf, obj_to_world_tsf=None):
    """visualization.py:83-107: every link's mesh in the robot (or, given `obj_to_world_tsf`, the world) frame under the
    current joint configuration -- a single one; a batch of configurations would need one mesh set per configuration.
    Links must be MeshSDF-like (an `obj_factory` with the loaded mesh; CachedSDF links reach it through `gt_sdf`)."""
    link_to_obj = tf.rigid_inverse(tf.as_matrix(robot_sdf.sdf.obj_frame_to_link_frame))
    if obj_to_world_tsf is not None:
        link_to_obj = tf.as_matrix(obj_to_world_tsf).to(link_to_obj) @ link_to_obj
    link_to_obj = link_to_obj.detach().cpu().numpy().astype(np.float64)
    S = len(robot_sdf.sdf_to_link_name)
    if link_to_obj.shape[0] != S:
        raise ValueError(f"{link_to_obj.shape[0]} link transforms for {S} links: set a single joint configuration first")
    meshes = []
    for i in range(S):
        leaf = robot_sdf.sdf.sdfs[i]
        factory = getattr(leaf, "obj_factory", None) or getattr(getattr(leaf, "gt_sdf", None), "obj_factory", None)
        if factory is None:
            raise TypeError(f"link {robot_sdf.sdf_to_link_name[i]}: no mesh behind a {type(leaf).__name__}")
        m = link_to_obj[i]
        meshes.append(factory._mesh.rotated(m[:3, :3]).translated(m[:3, 3]))
    return meshes
