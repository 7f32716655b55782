// CachedSDF query kernels (BASELINE config C2): nearest-voxel gather of the packed (val, grad) record with
// out-of-bounds handling, fused into one pass.  Replaces the ~25 stock torch kernels + 4 boolean-mask
// compactions of sdf.py:535-571 (reference).  HBM-bound: 12 B read + 16 B written per query point; the voxel
// grid (781 KB for the drill at 0.01 m) stays L2-resident, so streaming traffic uses non-temporal accesses.
#include "common.h"
#include "grid_lookup.h"

namespace pvamd {

// ---- vector path: one thread = 4 consecutive points = 3 x 16 B loads, 4 x 16 B stores ----
template <bool F64, bool WRITE_OOB>
__global__ __launch_bounds__(256) void cached_query_vec4(const pvamd_grid_t g, const f32x4* __restrict__ pts4,
                                                          int64_t ngroups, f32x4* __restrict__ val4,
                                                          f32x4* __restrict__ grad4, uint32_t* __restrict__ oob4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ngroups; i += stride) {
        const f32x4 a = __builtin_nontemporal_load(pts4 + 3 * i);
        const f32x4 b = __builtin_nontemporal_load(pts4 + 3 * i + 1);
        const f32x4 c = __builtin_nontemporal_load(pts4 + 3 * i + 2);
        bool v0, v1, v2, v3;
        const float4 r0 = cached_lookup<F64>(g, a.x, a.y, a.z, v0);
        const float4 r1 = cached_lookup<F64>(g, a.w, b.x, b.y, v1);
        const float4 r2 = cached_lookup<F64>(g, b.z, b.w, c.x, v2);
        const float4 r3 = cached_lookup<F64>(g, c.y, c.z, c.w, v3);
        __builtin_nontemporal_store(f32x4{r0.x, r1.x, r2.x, r3.x}, val4 + i);
        __builtin_nontemporal_store(f32x4{r0.y, r0.z, r0.w, r1.y}, grad4 + 3 * i);
        __builtin_nontemporal_store(f32x4{r1.z, r1.w, r2.y, r2.z}, grad4 + 3 * i + 1);
        __builtin_nontemporal_store(f32x4{r2.w, r3.y, r3.z, r3.w}, grad4 + 3 * i + 2);
        if constexpr (WRITE_OOB) {
            const uint32_t m = (v0 ? 0u : 1u) | (v1 ? 0u : 1u << 8) | (v2 ? 0u : 1u << 16) | (v3 ? 0u : 1u << 24);
            __builtin_nontemporal_store(m, oob4 + i);
        }
    }
}

// ---- scalar path: tail points and buffers that are not 16-byte aligned ----
template <bool F64>
__global__ __launch_bounds__(256) void cached_query_scalar(const pvamd_grid_t g, const float* __restrict__ pts,
                                                            int64_t first, int64_t P, float* __restrict__ val,
                                                            float* __restrict__ grad, uint8_t* __restrict__ oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        bool valid;
        const float4 r = cached_lookup<F64>(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], valid);
        val[i] = r.x;
        grad[3 * i] = r.y;
        grad[3 * i + 1] = r.z;
        grad[3 * i + 2] = r.w;
        if (oob) oob[i] = valid ? 0 : 1;
    }
}

template <bool F64>
__global__ __launch_bounds__(256) void cached_outside_kernel(const pvamd_grid_t g, const float* __restrict__ pts,
                                                              int64_t P, float level, uint8_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        int flat;
        const bool valid = voxel_flat<F64>(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], flat);
        // out-of-range points are assumed outside (sdf.py:599-601)
        out[i] = valid ? (uint8_t)(g.vox[4 * (int64_t)flat] > level) : (uint8_t)1;
    }
}

template <bool F64>
__global__ __launch_bounds__(256) void voxel_index_kernel(const pvamd_grid_t g, const float* __restrict__ pts,
                                                           int64_t P, int64_t* __restrict__ out_key,
                                                           int64_t* __restrict__ out_flat,
                                                           uint8_t* __restrict__ out_valid) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        long long key[3];
        const bool valid = voxel_key<F64>(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], key);
        if (out_key) {
            out_key[3 * i] = key[0];
            out_key[3 * i + 1] = key[1];
            out_key[3 * i + 2] = key[2];
        }
        // ravel exactly as the reference does: on the raw (unclamped) key, in int64
        if (out_flat) out_flat[i] = (key[0] * g.shape[1] + key[1]) * g.shape[2] + key[2];
        if (out_valid) out_valid[i] = valid ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void pack_grid_kernel(const float* __restrict__ val, const float* __restrict__ grad,
                                                         int64_t n, float4* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        out[i] = make_float4(val[i], grad[3 * i], grad[3 * i + 1], grad[3 * i + 2]);
    }
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_pack_grid(const float* val, const float* grad, int64_t n, float* out, void* stream) {
    if (!val || !grad || !out) return PVAMD_E_NULL;
    if (n < 0) return PVAMD_E_SHAPE;
    if (!aligned_to(out, 16)) return PVAMD_E_ALIGN;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_grid_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, val, grad, n,
                       reinterpret_cast<float4*>(out));
    return (int)hipGetLastError();
}

extern "C" int pvamd_cached_query(const pvamd_grid_t* grid, const float* points, int64_t P, float* out_val,
                                  float* out_grad, uint8_t* out_oob, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;  // empty query: nothing to read or write (torch hands out NULL for empty tensors)
    if (!grid || !out_val || !out_grad || !points) return PVAMD_E_NULL;
    if (int e = check_grid(*grid)) return e;
    if (!aligned_to(points, 4) || !aligned_to(out_val, 4) || !aligned_to(out_grad, 4)) return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const bool f64 = grid->index_f64 != 0;
    const bool vec_ok = aligned_to(points, 16) && aligned_to(out_val, 16) && aligned_to(out_grad, 16) &&
                        (!out_oob || aligned_to(out_oob, 4));
    const int64_t ngroups = vec_ok ? P / 4 : 0;
    if (ngroups > 0) {
        const dim3 grid_dim(stream_grid(ngroups, 256)), block(256);
        const f32x4* p4 = reinterpret_cast<const f32x4*>(points);
        f32x4* v4 = reinterpret_cast<f32x4*>(out_val);
        f32x4* g4 = reinterpret_cast<f32x4*>(out_grad);
        uint32_t* o4 = reinterpret_cast<uint32_t*>(out_oob);
        if (f64) {
            if (out_oob) hipLaunchKernelGGL((cached_query_vec4<true, true>), grid_dim, block, 0, s, *grid, p4, ngroups, v4, g4, o4);
            else hipLaunchKernelGGL((cached_query_vec4<true, false>), grid_dim, block, 0, s, *grid, p4, ngroups, v4, g4, o4);
        } else {
            if (out_oob) hipLaunchKernelGGL((cached_query_vec4<false, true>), grid_dim, block, 0, s, *grid, p4, ngroups, v4, g4, o4);
            else hipLaunchKernelGGL((cached_query_vec4<false, false>), grid_dim, block, 0, s, *grid, p4, ngroups, v4, g4, o4);
        }
    }
    const int64_t first = ngroups * 4;
    if (first < P) {
        const dim3 grid_dim(stream_grid(P - first, 256)), block(256);
        if (f64) hipLaunchKernelGGL((cached_query_scalar<true>), grid_dim, block, 0, s, *grid, points, first, P, out_val, out_grad, out_oob);
        else hipLaunchKernelGGL((cached_query_scalar<false>), grid_dim, block, 0, s, *grid, points, first, P, out_val, out_grad, out_oob);
    }
    return (int)hipGetLastError();
}

extern "C" int pvamd_cached_outside(const pvamd_grid_t* grid, const float* points, int64_t P, float level,
                                    uint8_t* out, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grid || !out || !points) return PVAMD_E_NULL;
    if (int e = check_grid(*grid)) return e;
    const dim3 grid_dim(stream_grid(P, 256)), block(256);
    if (grid->index_f64) hipLaunchKernelGGL((cached_outside_kernel<true>), grid_dim, block, 0, (hipStream_t)stream, *grid, points, P, level, out);
    else hipLaunchKernelGGL((cached_outside_kernel<false>), grid_dim, block, 0, (hipStream_t)stream, *grid, points, P, level, out);
    return (int)hipGetLastError();
}

extern "C" int pvamd_voxel_index(const pvamd_grid_t* grid, const float* points, int64_t P, int64_t* out_key,
                                 int64_t* out_flat, uint8_t* out_valid, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grid || !points) return PVAMD_E_NULL;
    if (int e = check_grid(*grid, /*need_vox=*/false)) return e;
    const dim3 grid_dim(stream_grid(P, 256)), block(256);
    if (grid->index_f64) hipLaunchKernelGGL((voxel_index_kernel<true>), grid_dim, block, 0, (hipStream_t)stream, *grid, points, P, out_key, out_flat, out_valid);
    else hipLaunchKernelGGL((voxel_index_kernel<false>), grid_dim, block, 0, (hipStream_t)stream, *grid, points, P, out_key, out_flat, out_valid);
    return (int)hipGetLastError();
}
