// Spatial processing order of a point set in seven small launches: bounds, Morton cell counts, scan (3), scatter.
// Replaces `torch.argsort(morton keys)` (a ~10-launch rocprim radix sort, ~60 us for 10 k points and 0.14 ms for 262 k:
// more than the mesh query of BASELINE C1 itself) in front of the mesh kernels and of the bucketed composed path.
//
// The kernels only need points that are neighbours in space to be neighbours in processing order, not a total order:
// this is a COUNTING sort on the leading `bits` bits of the 30-bit Morton key (16^3 cells up to 16 k points -- one
// workgroup does it all in LDS -- 32^3 up to 64 k, 64^3 up to
// a million, 128^3 beyond: about one point per cell or fewer -- with 8 points per cell the 64-point groups of the mesh
// kernels are visibly less compact, C5 7.9 -> 8.8 ms).  Cells come out in Z order; the order of the points inside a cell is whatever the atomics make it -- it may
// differ from run to run, and no result depends on it (the mesh kernels and the composed kernel return the same bits
// for any processing order; tests/test_mesh_gpu.py, tests/test_robot_gpu.py).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"
#include "morton.h"
#include "order_small.h"

#ifdef PVAMD_ORDER_MORTON
#define PVAMD_ORDER_KEY(x, y, z, lo, hi, b) morton_key30(x, y, z, lo, hi)
#else
#define PVAMD_ORDER_KEY(x, y, z, lo, hi, b) hilbert_key30(x, y, z, lo, hi, b)
#endif

namespace pvamd {

// scratch layout (uint32 words): [0..5] bounds codes, [8 .. 8 + cells) cell counters / offsets, then P keys, then
// cells / 1024 block sums of the scan
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
    // a block whose bound does not move the box leaves it alone (a stale read only costs an atomic)
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const unsigned code = order_code(fminf(fminf(part[0][d], part[1][d]), fminf(part[2][d], part[3][d])));
        if (code < __hip_atomic_load(box + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(box + d, code);
    } else if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        const unsigned code = order_code(fmaxf(fmaxf(part[0][d], part[1][d]), fmaxf(part[2][d], part[3][d])));
        if (code > __hip_atomic_load(box + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(box + d, code);
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
    const unsigned key = PVAMD_ORDER_KEY(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], lo, hi, (30 - shift) / 3);
    const unsigned cells = 1u << (30 - shift);
    scratch[kBoxWords + cells + i] = key;
    atomicAdd(scratch + kBoxWords + (key >> shift), 1u);
}

// exclusive scan of `cells` values in place, one block of 1024 threads (cells a multiple of 1024; used on the <= 2048
// block sums, 1-2 values per thread)
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

// ---- three-kernel scan: block sums, scan of the block sums, per-block scan ----
__global__ __launch_bounds__(1024) void order_blocksum_kernel(const unsigned* __restrict__ counters,
                                                              unsigned* __restrict__ blocksum) {
    __shared__ unsigned part[16];
    unsigned v = counters[(int64_t)blockIdx.x * 1024 + threadIdx.x];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned s = 0;
        for (int k = 0; k < 16; ++k) s += part[k];
        blocksum[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(1024) void order_blockscan_kernel(unsigned* __restrict__ counters,
                                                               const unsigned* __restrict__ blockprefix) {
    __shared__ unsigned sh[1024];
    const int t = threadIdx.x;
    const unsigned c = counters[(int64_t)blockIdx.x * 1024 + t];
    sh[t] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned add = t >= off ? sh[t - off] : 0u;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    counters[(int64_t)blockIdx.x * 1024 + t] = blockprefix[blockIdx.x] + sh[t] - c;  // exclusive
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

// ---- 1.5 million points and more: keys + a library radix sort ----
// The counting sort's two passes of P random atomics over 2^21 counters (8 MB: they execute memory-side) take 0.28 ms for
// 2 M points and grow linearly; rocPRIM's radix sort of the same (cell, index) pairs over the 21 key bits takes about a
// third of that (profiles/r04_mesh_variants.txt, section 10).  It is stable: points of one cell come out in index order.
__global__ __launch_bounds__(256) void order_keys_kernel(const float* __restrict__ pts, int64_t P, const unsigned* __restrict__ box,
                                                         unsigned* __restrict__ keys, int* __restrict__ index, int bits_per_axis) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = order_decode(box[d]);
        hi[d] = order_decode(box[3 + d]);
    }
    keys[i] = PVAMD_ORDER_KEY(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], lo, hi, bits_per_axis) >> (30 - 3 * bits_per_axis);
    index[i] = (int)i;
}

__global__ __launch_bounds__(256) void order_gather_kernel(const float* __restrict__ pts, int64_t P, const int* __restrict__ order,
                                                           int* __restrict__ inv, float* __restrict__ sorted_pts) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const int64_t i = order[k];
    if (inv) inv[i] = (int)k;
    if (sorted_pts) {
        sorted_pts[3 * k] = pts[3 * i];
        sorted_pts[3 * k + 1] = pts[3 * i + 1];
        sorted_pts[3 * k + 2] = pts[3 * i + 2];
    }
}

__global__ __launch_bounds__(1024) void order_small_kernel(const float* __restrict__ pts, int P, int* __restrict__ order,
                                                           int* __restrict__ inv, float* __restrict__ sorted_pts) {
    order_small_block(pts, P, order, inv, sorted_pts);
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
    if (P <= 16384) {
        hipLaunchKernelGGL(order_small_kernel, dim3(1), dim3(1024), 0, s, points, (int)P, order_out, inv_out, sorted_points_out);
        return (int)hipGetLastError();
    }
    unsigned* w = reinterpret_cast<unsigned*>(scratch);
    if (P >= PVAMD_ORDER_LIBRARY_SORT_FROM) {
        // scratch: [8] bounds codes | keys [P] | index [P] | sorted keys [P] | the library's temporary storage
        const int64_t want = (P + 255) / 256;
        hipLaunchKernelGGL(order_init_kernel, dim3(1), dim3(256), 0, s, w, 0);
        hipLaunchKernelGGL(order_bounds_kernel, dim3(want < 512 ? (unsigned)want : 512u), dim3(256), 0, s, points, P, w);  // (2048 blocks: 30-52 us for 2 M points against 20 -- they all reach their six atomics at once)
        unsigned* keys = w + kBoxWords;
        int* index = reinterpret_cast<int*>(keys + P);
        unsigned* keys_sorted = reinterpret_cast<unsigned*>(index + P);
        void* temp = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(keys_sorted + P) + 255) & ~(uintptr_t)255);  // the library's alignment
        const int bits = PVAMD_MORTON_ORDER_BITS(P);
        hipLaunchKernelGGL(order_keys_kernel, dim3((unsigned)want), dim3(256), 0, s, points, P, w, keys, index, bits / 3);
        size_t need = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keys, keys_sorted, index, order_out, (size_t)P, 0u, (unsigned)bits, s);
        if (e != hipSuccess) return (int)e;
        if (need > (size_t)PVAMD_ORDER_LIBRARY_TEMP_BYTES(P)) return PVAMD_E_SHAPE;  // the header's bound no longer holds
        e = rocprim::radix_sort_pairs(temp, need, keys, keys_sorted, index, order_out, (size_t)P, 0u, (unsigned)bits, s);
        if (e != hipSuccess) return (int)e;
        if (inv_out || sorted_points_out)
            hipLaunchKernelGGL(order_gather_kernel, dim3((unsigned)want), dim3(256), 0, s, points, P, order_out, inv_out, sorted_points_out);
        return (int)hipGetLastError();
    }
#ifdef PVAMD_ORDER_BITS_OVERRIDE
    const int bits = PVAMD_ORDER_BITS_OVERRIDE < PVAMD_MORTON_ORDER_BITS(P) ? PVAMD_ORDER_BITS_OVERRIDE : PVAMD_MORTON_ORDER_BITS(P), shift = 30 - bits, cells = 1 << bits;
#else
    const int bits = PVAMD_MORTON_ORDER_BITS(P), shift = 30 - bits, cells = 1 << bits;
#endif
    hipLaunchKernelGGL(order_init_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, w, cells);
    const int64_t want = (P + 255) / 256;
    hipLaunchKernelGGL(order_bounds_kernel, dim3(want < 512 ? (unsigned)want : 512u), dim3(256), 0, s, points, P, w);  // (2048 blocks: 30-52 us for 2 M points against 20 -- they all reach their six atomics at once)
    hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)want), dim3(256), 0, s, points, P, w, shift);
    {   // exclusive scan of the cell counters: block sums, scan of the <= 2048 block sums, per-block scan.  (One block
        // walking 32 consecutive counters per thread took 51 us for 32768 cells: serial, uncoalesced.)
        unsigned* blocksum = w + kBoxWords + cells + P;
        const int nblocks = cells / 1024;
        hipLaunchKernelGGL(order_blocksum_kernel, dim3(nblocks), dim3(1024), 0, s, w + kBoxWords, blocksum);
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, s, blocksum, nblocks < 1024 ? 1024 : nblocks);
        hipLaunchKernelGGL(order_blockscan_kernel, dim3(nblocks), dim3(1024), 0, s, w + kBoxWords, blocksum);
    }
    hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)want), dim3(256), 0, s, points, P, w, shift, order_out, inv_out,
                       sorted_points_out);
    return (int)hipGetLastError();
}
