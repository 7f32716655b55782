// ComposedSDF / RobotSDF query kernel (BASELINE configs C3, C4): for every (configuration a, point p) walk the S
// leaves in registers -- 3x4 affine into the leaf frame, nearest-voxel gather of the packed (val, grad) record (or the
// bounding-box fallback), running first-minimum -- rotate the winning gradient back once, and write 16 B.
// Replaces sdf.py:392-433 of the reference (transform_points broadcast to (S*A, P, 3), a Python loop of S
// CachedSDF calls, cat, argmin, gather): none of those intermediates reaches HBM here.
//
// Leaf descriptors and the S transforms of configuration a are wave-uniform, so they are read through the scalar
// cache into SGPRs (no LDS round trip); blockIdx.y = a keeps them uniform for the whole block.
#include "common.h"
#include "grid_lookup.h"

namespace pvamd {

struct Best {
    float v, gx, gy, gz;  // gradient kept in the winning leaf's frame until the end
    int s;
};

template <bool ANY_F64>
PVAMD_DEV void visit_leaf(const pvamd_grid_t& g, const float* __restrict__ M, int s, float px, float py, float pz,
                          Best& best) {
    const float x = affine_row(M[0], M[1], M[2], M[3], px, py, pz);
    const float y = affine_row(M[4], M[5], M[6], M[7], px, py, pz);
    const float z = affine_row(M[8], M[9], M[10], M[11], px, py, pz);
    bool valid;
    float4 r;
    if (ANY_F64 && g.index_f64) r = cached_lookup<true>(g, x, y, z, valid);
    else r = cached_lookup<false>(g, x, y, z, valid);
    // torch.argmin semantics (sdf.py:421): first minimum wins, NaN counts as the minimum
    const bool take = (best.s < 0) || (r.x < best.v) || (r.x != r.x && best.v == best.v);
    if (take) {
        best.v = r.x;
        best.gx = r.y;
        best.gy = r.z;
        best.gz = r.w;
        best.s = s;
    }
}

// g_obj = R^T g_leaf with R the obj->leaf rotation (sdf.py:409 transform_normals by the inverse transform)
PVAMD_DEV void rotate_back(const float* __restrict__ M, const Best& b, float& ox, float& oy, float& oz) {
    ox = fmaf(M[8], b.gz, fmaf(M[4], b.gy, mul_rn(M[0], b.gx)));
    oy = fmaf(M[9], b.gz, fmaf(M[5], b.gy, mul_rn(M[1], b.gx)));
    oz = fmaf(M[10], b.gz, fmaf(M[6], b.gy, mul_rn(M[2], b.gx)));
}

template <bool ANY_F64>
__global__ __launch_bounds__(256) void composed_query_vec4(const pvamd_grid_t* __restrict__ grids, int S,
                                                            const float* __restrict__ tf, int A,
                                                            const f32x4* __restrict__ pts4, int64_t ngroups,
                                                            f32x4* __restrict__ val4, f32x4* __restrict__ grad4,
                                                            int4* __restrict__ leaf4) {
    const int a = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ngroups; i += stride) {
        const f32x4 pa = pts4[3 * i], pb = pts4[3 * i + 1], pc = pts4[3 * i + 2];  // re-read per a: L2-resident
        const float px[4] = {pa.x, pa.w, pb.z, pc.y};
        const float py[4] = {pa.y, pb.x, pb.w, pc.z};
        const float pz[4] = {pa.z, pb.y, pc.x, pc.w};
        Best best[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) best[k] = Best{0.f, 0.f, 0.f, 0.f, -1};
        for (int s = 0; s < S; ++s) {
            const float* M = tf + 16 * ((int64_t)s * A + a);
            const pvamd_grid_t& g = grids[s];
#pragma unroll
            for (int k = 0; k < 4; ++k) visit_leaf<ANY_F64>(g, M, s, px[k], py[k], pz[k], best[k]);
        }
        float gx[4], gy[4], gz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // per-lane winner: the matrix row reads below are vector (not scalar) loads, but hit L1/L2
            const float* M = tf + 16 * ((int64_t)best[k].s * A + a);
            rotate_back(M, best[k], gx[k], gy[k], gz[k]);
        }
        const int64_t o = (int64_t)a * ngroups + i;  // P == 4*ngroups on this path
        __builtin_nontemporal_store(f32x4{best[0].v, best[1].v, best[2].v, best[3].v}, val4 + o);
        __builtin_nontemporal_store(f32x4{gx[0], gy[0], gz[0], gx[1]}, grad4 + 3 * o);
        __builtin_nontemporal_store(f32x4{gy[1], gz[1], gx[2], gy[2]}, grad4 + 3 * o + 1);
        __builtin_nontemporal_store(f32x4{gz[2], gx[3], gy[3], gz[3]}, grad4 + 3 * o + 2);
        if (leaf4) leaf4[o] = make_int4(best[0].s, best[1].s, best[2].s, best[3].s);
    }
}

template <bool ANY_F64>
__global__ __launch_bounds__(256) void composed_query_scalar(const pvamd_grid_t* __restrict__ grids, int S,
                                                              const float* __restrict__ tf, int A,
                                                              const float* __restrict__ pts, int64_t P,
                                                              float* __restrict__ val, float* __restrict__ grad,
                                                              int* __restrict__ leaf) {
    const int a = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        Best best{0.f, 0.f, 0.f, 0.f, -1};
        for (int s = 0; s < S; ++s) {
            visit_leaf<ANY_F64>(grids[s], tf + 16 * ((int64_t)s * A + a), s, px, py, pz, best);
        }
        float gx, gy, gz;
        rotate_back(tf + 16 * ((int64_t)best.s * A + a), best, gx, gy, gz);
        const int64_t o = (int64_t)a * P + i;
        val[o] = best.v;
        grad[3 * o] = gx;
        grad[3 * o + 1] = gy;
        grad[3 * o + 2] = gz;
        if (leaf) leaf[o] = best.s;
    }
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_composed_query(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A,
                                    const float* points, int64_t P, float* out_val, float* out_grad,
                                    int32_t* out_leaf, void* stream) {
    if (S < 1 || A < 1 || A > 65535 || P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grids || !tf || !out_val || !out_grad || !points) return PVAMD_E_NULL;
    if (!aligned_to(grids, 8) || !aligned_to(tf, 4) || !aligned_to(points, 4)) return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const bool vec_ok = (P % 4 == 0) && aligned_to(points, 16) && aligned_to(out_val, 16) && aligned_to(out_grad, 16) &&
                        (!out_leaf || aligned_to(out_leaf, 16));
    // The leaf descriptors live in device memory; whether any of them asks for float64 index arithmetic is not
    // known host-side, so the kernels are built for the general case and test the (wave-uniform) flag per leaf.
    if (vec_ok) {
        const int64_t ngroups = P / 4;
        // 2-D grid: x covers the points (capped; grid-stride), y = configuration
        const int64_t need = (ngroups + 255) / 256;
        const int64_t cap = ((int64_t)kNumCU * kMaxBlocksPerCU + A - 1) / A;
        const unsigned gx = (unsigned)(need < cap ? need : (cap < 1 ? 1 : cap));
        hipLaunchKernelGGL((composed_query_vec4<true>), dim3(gx, A), dim3(256), 0, s, grids, S, tf, A,
                           reinterpret_cast<const f32x4*>(points), ngroups, reinterpret_cast<f32x4*>(out_val),
                           reinterpret_cast<f32x4*>(out_grad), reinterpret_cast<int4*>(out_leaf));
    } else {
        const int64_t need = (P + 255) / 256;
        const int64_t cap = ((int64_t)kNumCU * kMaxBlocksPerCU + A - 1) / A;
        const unsigned gx = (unsigned)(need < cap ? need : (cap < 1 ? 1 : cap));
        hipLaunchKernelGGL((composed_query_scalar<true>), dim3(gx, A), dim3(256), 0, s, grids, S, tf, A, points, P,
                           out_val, out_grad, out_leaf);
    }
    return (int)hipGetLastError();
}
