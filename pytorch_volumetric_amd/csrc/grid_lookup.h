// Device-side voxel index arithmetic and nearest-voxel lookup shared by the cached / composed / chamfer kernels.
// gfx950 only.  Restates (not copies) what the reference asks of its value-range view:
//   index  = round_half_even((p - min) / res) as integer      (sdf.py:537, TorchMultidimView.ensure_index_key)
//   flat   = (kx*ny + ky)*nz + kz                              (sdf.py:538, ravel_multi_index)
//   valid  = all_d(min_d <= p_d <= max_d)                      (sdf.py:540, get_valid_values)
// in float32 or float64 according to pvamd_grid_t::index_f64 (see include/pvamd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "exact_math.h"
#include "../../include/pvamd.h"

namespace pvamd {

// hipcc contracts a*b+c into fma by default; every expression whose rounding is part of the contract below is
// written with explicit fmaf / __f*_rn so that the CPU oracle can state the same sequence.
#define PVAMD_DEV __device__ __forceinline__

template <bool F64>
PVAMD_DEV bool voxel_index_1d(const pvamd_grid_t& g, int d, float p, long long& k) {
    if constexpr (F64) {
        const double pd = (double)p;
        const bool valid = (g.dmin[d] <= pd) && (pd <= g.dmax[d]);
        k = (long long)__builtin_rint((pd - g.dmin[d]) / g.dres[d]);
        return valid;
    } else {
        const bool valid = (g.fmin[d] <= p) && (p <= g.fmax[d]);
        k = (long long)__builtin_rintf(div_rn(sub_rn(p, g.fmin[d]), g.fres[d]));
        return valid;
    }
}

// Full key (as the reference would hold it, int64, possibly outside [0, shape)) + validity.
template <bool F64>
PVAMD_DEV bool voxel_key(const pvamd_grid_t& g, float x, float y, float z, long long key[3]) {
    const bool vx = voxel_index_1d<F64>(g, 0, x, key[0]);
    const bool vy = voxel_index_1d<F64>(g, 1, y, key[1]);
    const bool vz = voxel_index_1d<F64>(g, 2, z, key[2]);
    return vx & vy & vz;
}

// In-bounds flat index for the gather.  Valid points always land in [0, shape) by construction
// ((max-min)/res rounds to shape-1); the clamp only protects the load against a malformed descriptor.
template <bool F64>
PVAMD_DEV bool voxel_flat(const pvamd_grid_t& g, float x, float y, float z, int& flat) {
    long long key[3];
    const bool valid = voxel_key<F64>(g, x, y, z, key);
    int kx = (int)key[0], ky = (int)key[1], kz = (int)key[2];
    kx = min(max(kx, 0), g.shape[0] - 1);
    ky = min(max(ky, 0), g.shape[1] - 1);
    kz = min(max(kz, 0), g.shape[2] - 1);
    flat = (kx * g.shape[1] + ky) * g.shape[2] + kz;
    return valid;
}

// BOUNDING_BOX fallback (sdf.py:559-571): per component dmin = max(bbmin - p, 0), dmax = max(p - bbmax, 0),
// t = dmin + dmax, negated where dmin > 0; val = |t|, grad = t / |t|  (0/0 = NaN when the point is inside
// the box, exactly as the reference).
PVAMD_DEV float4 bounding_box_sdf(const pvamd_grid_t& g, float x, float y, float z) {
    const float p[3] = {x, y, z};
    float t[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float lo = sub_rn(g.bb_min[d], p[d]);
        const bool lo_active = lo > 0.f;
        lo = lo_active ? lo : 0.f;
        float hi = sub_rn(p[d], g.bb_max[d]);
        hi = (hi > 0.f) ? hi : 0.f;
        const float s = add_rn(lo, hi);
        t[d] = lo_active ? -s : s;
    }
    const float n2 = fmaf(t[2], t[2], fmaf(t[1], t[1], mul_rn(t[0], t[0])));
    const float n = sqrt_rn(n2);
    return make_float4(n, div_rn(t[0], n), div_rn(t[1], n), div_rn(t[2], n));
}

// (val, gx, gy, gz) for one point in the leaf frame; `valid` reports the range test.
template <bool F64>
PVAMD_DEV float4 cached_lookup(const pvamd_grid_t& g, float x, float y, float z, bool& valid) {
    int flat;
    valid = voxel_flat<F64>(g, x, y, z, flat);
    if (valid) {
        return reinterpret_cast<const float4*>(g.vox)[flat];
    }
    if (g.oob_mode == PVAMD_OOB_BOUNDING_BOX) {
        return bounding_box_sdf(g, x, y, z);
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);  // LOOKUP_GT_SDF: zeros (sdf.py:546-547), caller fills in
}

// x' = M p for a row-major 4x4 (column-vector convention), k-ordered fma chain -- the rounding sequence of an
// f32 MFMA / a bmm k-loop: ((m0*px (+) m1*py) (+) m2*pz) + m3.
PVAMD_DEV float affine_row(float m0, float m1, float m2, float m3, float px, float py, float pz) {
    return add_rn(fmaf(m2, pz, fmaf(m1, py, mul_rn(m0, px))), m3);
}

}  // namespace pvamd
