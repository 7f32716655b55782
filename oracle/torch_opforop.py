"""Op-for-op torch restatement of CachedSDF.__call__ (reference sdf.py:535-571) -- TEST INFRASTRUCTURE ONLY.

BASELINE.md's "B1": the same sequence of stock torch ops and intermediates the reference executes (index -> ravel ->
range mask -> two masked gather/scatters -> compaction of the out-of-range points -> bounding-box branch -> two masked
scatters), so that bench.py can time "what a reference user gets on the CPU" next to the fused C port.  The absent
third-party view (multidim_indexing) is replaced by the three expressions its call sites amount to.
"""
import torch


class CachedOpForOp:
    def __init__(self, val_grid, grad_flat, vmin, vmax, bb):
        self.raw_data = val_grid.reshape(-1)          # view.raw_data
        self.shape = tuple(val_grid.shape)
        self.voxels_grad = grad_flat                  # sdf.py:523
        self.vmin, self.vmax = vmin, vmax
        self.res = (vmax - vmin) / (torch.tensor(self.shape, device=vmin.device) - 1)
        self.bb = bb

    def to(self, device):
        """Same op sequence on another device (e.g. the stock torch-ROCm kernels on the MI355X itself)."""
        return CachedOpForOp(self.raw_data.reshape(self.shape).to(device), self.voxels_grad.to(device),
                             self.vmin.to(device), self.vmax.to(device), self.bb.to(device))

    def __call__(self, pts):
        keys = torch.round((pts - self.vmin) / self.res).to(torch.long)                    # :537
        ravelled = (keys[..., 0] * self.shape[1] + keys[..., 1]) * self.shape[2] + keys[..., 2]  # :538
        inbound = ((self.vmin <= pts) & (pts <= self.vmax)).all(dim=-1)                    # :540
        oob = ~inbound                                                                      # :541
        val = torch.zeros(ravelled.shape, dtype=pts.dtype, device=pts.device)               # :546
        grad = torch.zeros(keys.shape, dtype=pts.dtype, device=pts.device)                  # :547
        val[inbound] = self.raw_data[ravelled[inbound]]                                     # :549
        grad[inbound] = self.voxels_grad[ravelled[inbound]]                                 # :550
        p = pts[oob]                                                                        # :552
        bb = self.bb.to(pts.dtype)
        dmin = bb[:, 0] - p                                                                 # :559
        dmin_active = dmin > 0
        dmin[~dmin_active] = 0
        dmax = p - bb[:, 1]                                                                 # :562
        dmax_active = dmax > 0
        dmax[~dmax_active] = 0
        dtotal = dmin + dmax                                                                # :565
        dtotal[dmin_active] = -dtotal[dmin_active]                                          # :567
        dist = dtotal.norm(dim=-1)                                                          # :568
        grad[oob] = dtotal / dist.unsqueeze(-1)                                             # :570
        val[oob] = dist                                                                     # :571
        return val, grad


class ComposedOpForOp:
    """ComposedSDF.__call__ (reference sdf.py:392-433) op for op: broadcast transform of all points into every leaf
    frame (materialising (S*A, P, 3)), a Python loop of per-leaf CachedSDF calls, gradient rotation by the inverse
    transforms, cat, argmin over leaves, gather.  TEST INFRASTRUCTURE ONLY (timed next to the fused kernel)."""

    def __init__(self, leaves, obj_to_leaf, batch=None):
        self.leaves = leaves                     # list of CachedOpForOp
        self.m = obj_to_leaf                     # (S*A, 4, 4), leaf-major
        self.batch = batch                       # A or None
        self.m_inv = torch.linalg.inv(obj_to_leaf)  # :380

    def __call__(self, points):
        pts_shape = points.shape
        p = points.reshape(-1, 3)                                                        # :395
        S = len(self.leaves)
        A = self.batch or 1
        x = p.unsqueeze(0) @ self.m[:, :3, :3].transpose(-1, -2) + self.m[:, None, :3, 3]  # :399 (S*A, P, 3)
        x = x.reshape(S, A, *p.shape)                                                    # :402
        vs, gs = [], []
        for i, leaf in enumerate(self.leaves):                                           # :405
            v, g = leaf(x[i])                                                            # :407
            rot = self.m_inv[i * A:(i + 1) * A, :3, :3]
            g = g @ rot.transpose(-1, -2)                                                # :409 transform_normals
            vs.append(v)
            gs.append(g)
        v = torch.cat(vs).reshape(S, -1)                                                 # :414-418
        g = torch.cat(gs).reshape(S, -1, 3)
        closest = torch.argmin(v, 0)                                                     # :421
        idx = torch.arange(0, v.shape[1], device=v.device)
        vv, gg = v[closest, idx], g[closest, idx]                                        # :425-426
        if self.batch is not None:
            vv = vv.reshape(A, *pts_shape[:-1])
            gg = gg.reshape(A, *pts_shape[:-1], 3)
        return vv, gg
