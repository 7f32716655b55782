// Seeded area-uniform surface sampling on the device: the draw behind sample_mesh_points (reference sdf.py:643-650,
// open3d's mesh.sample_points_uniformly followed by a random subset).  open3d's generator cannot be reproduced, so
// the draw is counter-based (splitmix64 of (seed, sample index)) -- identical on host and device, independent of the
// launch geometry -- and keeps open3d's construction: pick a triangle with probability proportional to its area,
// then p = (1 - sqrt(r1)) a + sqrt(r1) (1 - r2) b + sqrt(r1) r2 c.
#include "common.h"
#include "mesh_math.h"

namespace pvamd {

PVAMD_DEV double uniform01(uint64_t seed, int64_t index, int k) {
    const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)index * 4u + (uint64_t)k));
    return (double)(h >> 11) * 0x1.0p-53;  // 53 random bits in [0, 1)
}

__global__ __launch_bounds__(256) void sample_surface_kernel(const float* __restrict__ tri, const double* __restrict__ cdf,
                                                              int F, int64_t n, uint64_t seed,
                                                              double* __restrict__ out_points, int* __restrict__ out_face,
                                                              int64_t* __restrict__ out_key) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double u = uniform01(seed, i, 0);
        // first triangle whose cumulative area fraction exceeds u (upper bound); the last one catches round-off
        int lo = 0, hi = F - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > u) hi = mid;
            else lo = mid + 1;
        }
        const float* t = tri + 9 * (int64_t)lo;
        const double s = __builtin_sqrt(uniform01(seed, i, 1)), r2 = uniform01(seed, i, 2);
        const double wa = 1.0 - s, wb = s * (1.0 - r2), wc = s * r2;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            out_points[3 * i + d] = __builtin_fma(wc, (double)t[6 + d], __builtin_fma(wb, (double)t[3 + d], wa * (double)t[d]));
        }
        if (out_face) out_face[i] = lo;
        if (out_key) out_key[i] = (int64_t)(splitmix64(seed ^ splitmix64((uint64_t)i * 4u + 3u)) >> 1);  // subset-selection key
    }
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_sample_surface(const float* tri, const double* cdf, int32_t F, int64_t n, uint64_t seed,
                                    double* out_points, int32_t* out_face, int64_t* out_key, void* stream) {
    if (F < 1 || n < 0) return PVAMD_E_SHAPE;
    if (n == 0) return 0;
    if (!tri || !cdf || !out_points) return PVAMD_E_NULL;
    if (!aligned_to(cdf, 8) || !aligned_to(out_points, 8) || !aligned_to(tri, 4)) return PVAMD_E_ALIGN;
    hipLaunchKernelGGL(sample_surface_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, tri, cdf, F, n,
                       seed, out_points, out_face, out_key);
    return (int)hipGetLastError();
}
