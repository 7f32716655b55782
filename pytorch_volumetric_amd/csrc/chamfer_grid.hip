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
