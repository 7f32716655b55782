// batch_chamfer_dist against a cached voxel grid (obj_sdf branch, reference chamfer.py:84-85,92-94): transform the
// points by each of the B world->object matrices, nearest-voxel lookup of the value, and reduce sum (scale*v)^2
// per transform.  The (B, N, 3) transformed cloud and the (B, N) distances of the reference never reach HBM.
#include "common.h"
#include "grid_lookup.h"

namespace pvamd {

template <bool F64>
__global__ __launch_bounds__(256) void chamfer_grid_kernel(const pvamd_grid_t g, const float* __restrict__ W,
                                                            const float* __restrict__ pts, int64_t N, float scale,
                                                            double* __restrict__ out_sum) {
    __shared__ double scratch[4];
    const float* M = W + 16 * (int64_t)blockIdx.y;  // wave-uniform: scalar loads
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        const float x = affine_row(M[0], M[1], M[2], M[3], px, py, pz);
        const float y = affine_row(M[4], M[5], M[6], M[7], px, py, pz);
        const float z = affine_row(M[8], M[9], M[10], M[11], px, py, pz);
        bool valid;
        const float4 r = cached_lookup<F64>(g, x, y, z, valid);
        const float sd = mul_rn(scale, r.x);
        acc += (double)mul_rn(sd, sd);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += scratch[w];
        atomicAdd(out_sum + blockIdx.y, total);
    }
}

// ---- PlausibleDiversity's reduction (chamfer.py:185-195) in one pass over the (B, P) chamfer matrix ----
// block b < B: the minimum of row b (first index on ties, NaN counts as the minimum -- torch.min);
// block B + c: the minima of columns 256 c .. 256 c + 255 (thread = column: coalesced across a row).
template <typename T>
__global__ __launch_bounds__(256) void pairwise_min_kernel(const T* __restrict__ E, int B, int P, T* __restrict__ row_val,
                                                          int64_t* __restrict__ row_idx, T* __restrict__ col_val,
                                                          int64_t* __restrict__ col_idx) {
    auto better = [](T v, int i, T bv, int bi) {  // (v, i) before (bv, bi): smaller value, NaN smallest, then lower index
        const bool vn = v != v, bn = bv != bv;
        if (vn != bn) return vn;
        if (!vn && v != bv) return v < bv;
        return i < bi;
    };
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        T v = E[(int64_t)b * P + (threadIdx.x < (unsigned)P ? threadIdx.x : 0)];
        int idx = threadIdx.x < (unsigned)P ? (int)threadIdx.x : 0;
        for (int j = threadIdx.x + 256; j < P; j += 256) {
            const T w = E[(int64_t)b * P + j];
            if (better(w, j, v, idx)) { v = w; idx = j; }
        }
        __shared__ T sv[256];
        __shared__ int si[256];
        sv[threadIdx.x] = v; si[threadIdx.x] = idx;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off && better(sv[threadIdx.x + off], si[threadIdx.x + off], sv[threadIdx.x], si[threadIdx.x])) {
                sv[threadIdx.x] = sv[threadIdx.x + off];
                si[threadIdx.x] = si[threadIdx.x + off];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) { row_val[b] = sv[0]; row_idx[b] = si[0]; }
    } else {
        const int c = ((int)blockIdx.x - B) * 256 + threadIdx.x;
        if (c >= P) return;
        T v = E[c];
        int idx = 0;
        for (int b = 1; b < B; ++b) {
            const T w = E[(int64_t)b * P + c];
            if (better(w, b, v, idx)) { v = w; idx = b; }
        }
        col_val[c] = v; col_idx[c] = idx;
    }
}

// sums[0] = mean of the row minima, sums[1] = mean of the column minima: one wave, fixed order, float64
template <typename T>
__global__ __launch_bounds__(64) void pairwise_mean_kernel(const T* __restrict__ row_val, int B, const T* __restrict__ col_val, int P,
                                                          double* __restrict__ sums) {
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < B; i += 64) a += (double)row_val[i];
    for (int i = threadIdx.x; i < P; i += 64) c += (double)col_val[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        c += __shfl_down(c, off, 64);
    }
    if (threadIdx.x == 0) { sums[0] = a / B; sums[1] = c / P; }
}

__global__ void zero_f64_kernel2(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_chamfer_grid(const pvamd_grid_t* grid, const float* W, int32_t B, const float* points, int64_t N,
                                  float scale, double* out_sum, void* stream) {
    if (!grid || !out_sum) return PVAMD_E_NULL;
    if (B < 0 || N < 0) return PVAMD_E_SHAPE;
    if (int e = check_grid(*grid)) return e;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_f64_kernel2, dim3((B + 255) / 256), dim3(256), 0, s, out_sum, B);
    if (N == 0) return (int)hipGetLastError();
    if (!W || !points) return PVAMD_E_NULL;
    const int64_t need = (N + 255) / 256;
    for (int32_t b0 = 0; b0 < B; b0 += 65535) {
        const int32_t nb = (B - b0) < 65535 ? (B - b0) : 65535;
        const int64_t cap = ((int64_t)kNumCU * kMaxBlocksPerCU + nb - 1) / nb;
        const unsigned gx = (unsigned)(need < cap ? need : (cap < 1 ? 1 : cap));
        if (grid->index_f64) hipLaunchKernelGGL((chamfer_grid_kernel<true>), dim3(gx, nb), dim3(256), 0, s, *grid, W + 16 * (int64_t)b0, points, N, scale, out_sum + b0);
        else hipLaunchKernelGGL((chamfer_grid_kernel<false>), dim3(gx, nb), dim3(256), 0, s, *grid, W + 16 * (int64_t)b0, points, N, scale, out_sum + b0);
    }
    return (int)hipGetLastError();
}

extern "C" int pvamd_pairwise_min_reduce(const void* errors, int32_t is_f64, int32_t B, int32_t P, void* row_val, int64_t* row_idx,
                                         void* col_val, int64_t* col_idx, double* means, void* stream) {
    if (!errors || !row_val || !row_idx || !col_val || !col_idx || !means) return PVAMD_E_NULL;
    if (B < 1 || P < 1) return PVAMD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)B + (unsigned)((P + 255) / 256);
    if (is_f64) {
        hipLaunchKernelGGL((pairwise_min_kernel<double>), dim3(blocks), dim3(256), 0, s, (const double*)errors, B, P, (double*)row_val,
                           row_idx, (double*)col_val, col_idx);
        hipLaunchKernelGGL((pairwise_mean_kernel<double>), dim3(1), dim3(64), 0, s, (const double*)row_val, B, (const double*)col_val, P, means);
    } else {
        hipLaunchKernelGGL((pairwise_min_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)errors, B, P, (float*)row_val,
                           row_idx, (float*)col_val, col_idx);
        hipLaunchKernelGGL((pairwise_mean_kernel<float>), dim3(1), dim3(64), 0, s, (const float*)row_val, B, (const float*)col_val, P, means);
    }
    return (int)hipGetLastError();
}
