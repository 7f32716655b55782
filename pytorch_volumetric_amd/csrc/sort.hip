// Spatial processing order of a point set: up to 786,432 points in seven small launches (bounds, curve cell counts, scan (3),
// scatter); beyond, an LSD radix sort of (cell, index) pairs (below).  All hand-written: no library call on this path.
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

// ---- 786,432 points and more: an LSD radix sort of (cell, index) pairs ----
// The counting sort's two passes of P random atomics over 2^21 counters (8 MB: they execute memory-side) take 0.28 ms for
// 2 M points and grow linearly.  Here the 21 key bits go in three stable passes of 7 bits over 4096-pair tiles:
//   histogram   one block per tile counts its 128 digits in LDS -> table[tile][digit], and adds them to the totals of its
//               SUPERGROUP of G ~ sqrt(tiles) consecutive tiles (a few atomics per block; pass 1 computes the keys as well)
//   scatter     one block per tile: where its keys of digit d go = (keys of smaller digits anywhere) + (digit d in earlier
//               supergroups) + (digit d in earlier tiles of its own supergroup) -- 2 sqrt(tiles) coalesced, L2-resident
//               reads per thread instead of a scan launch over the table; ranks inside the tile are STABLE (per wave,
//               64 keys a round: the lanes that hold the same digit find each other with 7 ballots, the lowest of them
//               bumps the wave's digit counter); the tile is reordered in LDS and leaves in runs of consecutive addresses
// Stable throughout: the points of one cell come out in index order, the same permutation on every run.  Round 4 called
// rocprim::radix_sort_pairs here (3 x 23.6 us + 4.9 us at 2 M pairs; 1.9 MB of template instantiations in the .so).
constexpr int kRadixBits = 7;
constexpr int kRadixBins = 1 << kRadixBits;
constexpr int kRadixThreads = 256;
constexpr int kRadixRounds = 16;                          // keys per thread
constexpr int kRadixTile = kRadixThreads * kRadixRounds;  // 4096 pairs per block
constexpr int kRadixMaxPasses = 3;
static_assert(PVAMD_MORTON_ORDER_BITS(1ll << 40) <= kRadixBits * kRadixMaxPasses, "more key bits than radix passes");

// table[tile][digit] and the supergroup totals from this block's LDS histogram
PVAMD_DEV void radix_publish_histogram(const unsigned* hist, unsigned* __restrict__ table, unsigned* __restrict__ sgtotal, int G) {
    if (threadIdx.x < kRadixBins) {
        const unsigned c = hist[threadIdx.x];
        table[(int64_t)blockIdx.x * kRadixBins + threadIdx.x] = c;
        if (c) atomicAdd(sgtotal + (int64_t)(blockIdx.x / G) * kRadixBins + threadIdx.x, c);
    }
}

// pass 1: the keys (cell along the curve) of one tile and the histogram of their lowest digit
__global__ __launch_bounds__(kRadixThreads) void radix_keys_hist_kernel(const float* __restrict__ pts, int64_t P,
                                                                         const unsigned* __restrict__ box,
                                                                         unsigned* __restrict__ keys, int bits_per_axis,
                                                                         unsigned* __restrict__ table,
                                                                         unsigned* __restrict__ sgtotal, int G) {
    __shared__ unsigned hist[kRadixBins];
    if (threadIdx.x < kRadixBins) hist[threadIdx.x] = 0u;
    float lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = order_decode(box[d]);
        hi[d] = order_decode(box[3 + d]);
    }
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * kRadixTile;
    for (int r = 0; r < kRadixRounds; ++r) {
        const int64_t i = t0 + r * kRadixThreads + threadIdx.x;
        if (i < P) {
            const unsigned key = PVAMD_ORDER_KEY(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], lo, hi, bits_per_axis) >> (30 - 3 * bits_per_axis);
            keys[i] = key;
            atomicAdd(&hist[key & (kRadixBins - 1)], 1u);
        }
    }
    __syncthreads();
    radix_publish_histogram(hist, table, sgtotal, G);
}

__global__ __launch_bounds__(kRadixThreads) void radix_hist_kernel(const unsigned* __restrict__ keys, int64_t P, int shift,
                                                                    unsigned* __restrict__ table,
                                                                    unsigned* __restrict__ sgtotal, int G) {
    __shared__ unsigned hist[kRadixBins];
    if (threadIdx.x < kRadixBins) hist[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * kRadixTile;
#pragma unroll 4
    for (int r = 0; r < kRadixRounds; ++r) {
        const int64_t i = t0 + r * kRadixThreads + threadIdx.x;
        if (i < P) atomicAdd(&hist[(keys[i] >> shift) & (kRadixBins - 1)], 1u);
    }
    __syncthreads();
    radix_publish_histogram(hist, table, sgtotal, G);
}

// inclusive sum over the 64 lanes of a wave
PVAMD_DEV unsigned wave_inclusive_sum(unsigned v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(v, off, 64);
        if (lane >= off) v += up;
    }
    return v;
}

// FIRST: the index of a pair is its position (no index array yet).  LAST: only the indices leave (the order).
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kRadixThreads) void radix_scatter_kernel(const unsigned* __restrict__ keys_in,
                                                                       const int* __restrict__ idx_in, int64_t P, int shift,
                                                                       const unsigned* __restrict__ table,
                                                                       const unsigned* __restrict__ sgtotal, int G, int nsg,
                                                                       unsigned* __restrict__ keys_out, int* __restrict__ idx_out) {
    __shared__ unsigned wave_hist[kRadixThreads / 64][kRadixBins];  // a wave's running digit counts, then its offset inside the tile's digit run
    __shared__ unsigned base[kRadixBins];        // global position of the tile's first key of digit d, minus tile_start[d]
    __shared__ unsigned tile_start[kRadixBins];  // where digit d starts inside the reordered tile
    __shared__ unsigned carry[2];
    __shared__ unsigned skeys[kRadixTile];
    __shared__ int sidx[kRadixTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, sg = tile / G;
    const int64_t t0 = (int64_t)tile * kRadixTile;
    const int n = (int)(P - t0 < kRadixTile ? P - t0 : kRadixTile);
    for (int k = tid; k < (kRadixThreads / 64) * kRadixBins; k += kRadixThreads) (&wave_hist[0][0])[k] = 0u;
    // phase 0: where this tile's keys of digit d (= tid) start in the output
    unsigned tot = 0, pre = 0;
    if (tid < kRadixBins) {
        for (int s2 = 0; s2 < nsg; ++s2) {
            const unsigned c = sgtotal[(int64_t)s2 * kRadixBins + tid];
            tot += c;
            pre += s2 < sg ? c : 0u;
        }
        for (int t = sg * G; t < tile; ++t) pre += table[(int64_t)t * kRadixBins + tid];
    }
    const unsigned inc = wave_inclusive_sum(tot, lane);
    if (tid == 63) carry[0] = inc;
    __syncthreads();
    if (tid < kRadixBins) base[tid] = inc - tot + (tid >= 64 ? carry[0] : 0u) + pre;
    // phase 1: stable rank of every key among the keys of its digit in its wave's 1024-key slice (round by round)
    unsigned key[kRadixRounds];
    int idx[kRadixRounds];
    unsigned rank[kRadixRounds];
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kRadixRounds; ++r) {
        const int e = wave * (64 * kRadixRounds) + r * 64 + lane;
        const bool live = e < n;
        key[r] = live ? keys_in[t0 + e] : 0xffffffffu;
        if (!FIRST) idx[r] = live ? idx_in[t0 + e] : 0;
        const unsigned d = (key[r] >> shift) & (kRadixBins - 1);
        uint64_t same = __builtin_amdgcn_ballot_w64(live);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t with = __builtin_amdgcn_ballot_w64(bit);
            same &= bit ? with : ~with;
        }
        const unsigned before = (unsigned)__popcll(same & below);
        const unsigned old = wave_hist[wave][d];
        PVAMD_WAVE_SYNC();  // every lane has read the counter before the digit's first lane moves it
        if (live && before == 0) wave_hist[wave][d] = old + (unsigned)__popcll(same);
        PVAMD_WAVE_SYNC();
        rank[r] = old + before;
    }
    __syncthreads();
    // phase 2: the waves' counts per digit -> each wave's offset inside the digit's run; the runs' starts inside the tile
    unsigned cnt = 0;
    if (tid < kRadixBins) {
        unsigned run = 0;
#pragma unroll
        for (int w2 = 0; w2 < kRadixThreads / 64; ++w2) {
            const unsigned c = wave_hist[w2][tid];
            wave_hist[w2][tid] = run;
            run += c;
        }
        cnt = run;
    }
    const unsigned inc2 = wave_inclusive_sum(cnt, lane);
    if (tid == 63) carry[1] = inc2;
    __syncthreads();
    if (tid < kRadixBins) {
        const unsigned start = inc2 - cnt + (tid >= 64 ? carry[1] : 0u);
        tile_start[tid] = start;
        base[tid] -= start;  // (unsigned wrap-around is fine: base[d] + position inside the tile is the global position)
    }
    __syncthreads();
    // phase 3: the tile in digit order, in LDS
#pragma unroll
    for (int r = 0; r < kRadixRounds; ++r) {
        const int e = wave * (64 * kRadixRounds) + r * 64 + lane;
        if (e < n) {
            const unsigned d = (key[r] >> shift) & (kRadixBins - 1);
            const unsigned at = tile_start[d] + wave_hist[wave][d] + rank[r];
            skeys[at] = key[r];
            sidx[at] = FIRST ? (int)(t0 + e) : idx[r];
        }
    }
    __syncthreads();
    // phase 4: out, in runs of consecutive addresses (one run per digit)
    for (int i = tid; i < n; i += kRadixThreads) {
        const unsigned k = skeys[i];
        const unsigned g = base[(k >> shift) & (kRadixBins - 1)] + (unsigned)i;
        if (!LAST) keys_out[g] = k;
        idx_out[g] = sidx[i];
    }
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
    if (P >= PVAMD_ORDER_RADIX_SORT_FROM) {
        // scratch (words): [8] bounds codes | supergroup totals [3][nsg][128] | table [tiles][128] | keys A [P] | keys B [P] |
        // index A [P] | index B [P]
        const int64_t want = (P + 255) / 256;
        const int64_t tiles = (P + kRadixTile - 1) / kRadixTile;
        int G = 1;
        while ((int64_t)G * G < tiles) ++G;  // ~sqrt(tiles) tiles per supergroup
        const int nsg = (int)((tiles + G - 1) / G);
        const int bits = PVAMD_MORTON_ORDER_BITS(P);
        const int passes = (bits + kRadixBits - 1) / kRadixBits;
        unsigned* sgtotal = w + kBoxWords;
        unsigned* table = sgtotal + (int64_t)kRadixMaxPasses * nsg * kRadixBins;
        unsigned* keys[2] = {table + tiles * kRadixBins, table + tiles * kRadixBins + P};
        int* index[2] = {reinterpret_cast<int*>(keys[1] + P), reinterpret_cast<int*>(keys[1] + 2 * P)};
        const int zero_words = kRadixMaxPasses * nsg * kRadixBins;
        hipLaunchKernelGGL(order_init_kernel, dim3((zero_words + 255) / 256), dim3(256), 0, s, w, zero_words);
        hipLaunchKernelGGL(order_bounds_kernel, dim3(want < 512 ? (unsigned)want : 512u), dim3(256), 0, s, points, P, w);  // (2048 blocks: 30-52 us for 2 M points against 20 -- they all reach their six atomics at once)
        for (int p = 0; p < passes; ++p) {
            unsigned* sgt = sgtotal + (int64_t)p * nsg * kRadixBins;
            const unsigned* kin = keys[p & 1];
            unsigned* kout = keys[(p + 1) & 1];
            const int* iin = index[p & 1];
            int* iout = p == passes - 1 ? order_out : index[(p + 1) & 1];
            const int shift = p * kRadixBits;
            if (p == 0)
                hipLaunchKernelGGL(radix_keys_hist_kernel, dim3((unsigned)tiles), dim3(kRadixThreads), 0, s, points, P, w, keys[0], bits / 3,
                                   table, sgt, G);
            else
                hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)tiles), dim3(kRadixThreads), 0, s, kin, P, shift, table, sgt, G);
            const bool first = p == 0, last = p == passes - 1;
            if (first && last)
                hipLaunchKernelGGL((radix_scatter_kernel<true, true>), dim3((unsigned)tiles), dim3(kRadixThreads), 0, s, kin, iin, P, shift, table, sgt, G, nsg, kout, iout);
            else if (first)
                hipLaunchKernelGGL((radix_scatter_kernel<true, false>), dim3((unsigned)tiles), dim3(kRadixThreads), 0, s, kin, iin, P, shift, table, sgt, G, nsg, kout, iout);
            else if (last)
                hipLaunchKernelGGL((radix_scatter_kernel<false, true>), dim3((unsigned)tiles), dim3(kRadixThreads), 0, s, kin, iin, P, shift, table, sgt, G, nsg, kout, iout);
            else
                hipLaunchKernelGGL((radix_scatter_kernel<false, false>), dim3((unsigned)tiles), dim3(kRadixThreads), 0, s, kin, iin, P, shift, table, sgt, G, nsg, kout, iout);
        }
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
