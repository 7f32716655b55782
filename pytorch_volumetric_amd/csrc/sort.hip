// Spatial processing order of a point set in five small launches: bounds, Morton cell counts, scan, scatter.
// Replaces `torch.argsort(morton keys)` (a ~10-launch rocprim radix sort, ~60 us for 10 k points and 0.14 ms for 262 k:
// more than the mesh query of BASELINE C1 itself) in front of the mesh kernels and of the bucketed composed path.
//
// The kernels only need points that are neighbours in space to be neighbours in processing order, not a total order:
// this is a COUNTING sort on the leading `bits` bits of the 30-bit Morton key (32^3 cells, 64^3 beyond a million
// points).  Cells come out in Z order; the order of the points inside a cell is whatever the atomics make it -- it may
// differ from run to run, and no result depends on it (the mesh kernels and the composed kernel return the same bits
// for any processing order; tests/test_mesh_gpu.py, tests/test_robot_gpu.py).
#include "common.h"
#include "morton.h"

namespace pvamd {

// scratch layout (uint32 words): [0..5] bounds codes, [6] unused, [8 .. 8 + cells) cell counters / offsets,
// then P keys
constexpr int kBoxWords = 8;

__global__ __launch_bounds__(256) void order_init_kernel(unsigned* __restrict__ scratch, int cells) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3) scratch[i] = order_code(INFINITY);
    else if (i < 6) scratch[i] = order_code(-INFINITY);
    if (i < cells) scratch[kBoxWords + i] = 0u;
}

__global__ __launch_bounds__(256) void order_bounds_kernel(const float* __restrict__ pts, int64_t P, unsigned* box) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = pts[3 * i + d];
            if (fabsf(v) < INFINITY) {  // false for NaN and +-inf
                lo[d] = fminf(lo[d], v);
                hi[d] = fmaxf(hi[d], v);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
    }
    __shared__ float part[4][6];  // one set of atomics per block (see aabb_reduce_kernel in mesh.hip)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { part[wave][d] = lo[d]; part[wave][3 + d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        atomicMin(box + d, order_code(fminf(fminf(part[0][d], part[1][d]), fminf(part[2][d], part[3][d]))));
    } else if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        atomicMax(box + d, order_code(fmaxf(fmaxf(part[0][d], part[1][d]), fmaxf(part[2][d], part[3][d]))));
    }
}

__global__ __launch_bounds__(256) void order_count_kernel(const float* __restrict__ pts, int64_t P,
                                                          unsigned* __restrict__ scratch, int shift) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = order_decode(scratch[d]);
        hi[d] = order_decode(scratch[3 + d]);
    }
    const unsigned key = morton_key30(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], lo, hi);
    const unsigned cells = 1u << (30 - shift);
    scratch[kBoxWords + cells + i] = key;
    atomicAdd(scratch + kBoxWords + (key >> shift), 1u);
}

// exclusive scan of `cells` counters in place, one block of 1024 threads (cells is a multiple of 1024)
__global__ __launch_bounds__(1024) void order_scan_kernel(unsigned* __restrict__ counters, int cells) {
    __shared__ unsigned partial[1024];
    const int per = cells / 1024, t = threadIdx.x;
    unsigned sum = 0;
    for (int k = 0; k < per; ++k) sum += counters[t * per + k];
    partial[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        const unsigned add = t >= off ? partial[t - off] : 0u;
        __syncthreads();
        partial[t] += add;
        __syncthreads();
    }
    unsigned run = partial[t] - sum;  // exclusive prefix of this thread's chunk
    for (int k = 0; k < per; ++k) {
        const unsigned c = counters[t * per + k];
        counters[t * per + k] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void order_scatter_kernel(const float* __restrict__ pts, int64_t P,
                                                            unsigned* __restrict__ scratch, int shift,
                                                            int* __restrict__ order, int* __restrict__ inv,
                                                            float* __restrict__ sorted_pts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const unsigned cells = 1u << (30 - shift);
    const unsigned key = scratch[kBoxWords + cells + i];
    const unsigned slot = atomicAdd(scratch + kBoxWords + (key >> shift), 1u);
    order[slot] = (int)i;
    if (inv) inv[i] = (int)slot;
    if (sorted_pts) {
        sorted_pts[3 * (int64_t)slot] = pts[3 * i];
        sorted_pts[3 * (int64_t)slot + 1] = pts[3 * i + 1];
        sorted_pts[3 * (int64_t)slot + 2] = pts[3 * i + 2];
    }
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_morton_order(const float* points, int64_t P, int32_t* order_out, int32_t* inv_out,
                                  float* sorted_points_out, void* scratch, void* stream) {
    if (P < 0 || P > 0x7fffffffLL) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!points || !order_out || !scratch) return PVAMD_E_NULL;
    if (!aligned_to(scratch, 4) || !aligned_to(points, 4)) return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    unsigned* w = reinterpret_cast<unsigned*>(scratch);
    const int bits = PVAMD_MORTON_ORDER_BITS(P), shift = 30 - bits, cells = 1 << bits;
    hipLaunchKernelGGL(order_init_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, w, cells);
    const int64_t want = (P + 255) / 256;
    hipLaunchKernelGGL(order_bounds_kernel, dim3(want < 512 ? (unsigned)want : 512u), dim3(256), 0, s, points, P, w);
    hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)want), dim3(256), 0, s, points, P, w, shift);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, s, w + kBoxWords, cells);
    hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)want), dim3(256), 0, s, points, P, w, shift, order_out, inv_out,
                       sorted_points_out);
    return (int)hipGetLastError();
}
