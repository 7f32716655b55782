// CachedSDF query kernels (BASELINE config C2): nearest-voxel gather of the packed (val, grad) record with
// out-of-bounds handling, fused into one pass.  Replaces the ~25 stock torch kernels + 4 boolean-mask
// compactions of sdf.py:535-571 (reference).  HBM-bound: 12 B read + 16 B written per query point; the voxel
// grid (781 KB for the drill at 0.01 m) stays L2-resident, so streaming traffic uses non-temporal accesses.
#include "common.h"
#include "grid_lookup.h"

namespace pvamd {

// ---- wave-tile path: one wave = 256 consecutive points per pass ----
// Every global instruction moves a contiguous 1 KB (64 lanes x 16 B): 3 loads bring the tile's 768 floats of xyz into
// a wave-private LDS slice; the results go back through the same slice and leave as 1 + 3 contiguous 1 KB stores.
// No block barrier: a wave's LDS traffic is ordered.  Measured against the previous "4 points per thread, 48-byte
// strided float4" form: 0.63 ms -> 0.38 ms for 64M points (tools/kbench.hip, profiles/r01_kbench.txt).
// Which points a lane owns (PVAMD_CQ_OWN4):
//   0  lane, lane+64, lane+128, lane+192: stride-3 dword LDS reads (conflict-free), 16 dword LDS writes for the results
//   1  4*lane .. 4*lane+3: the lane's 12 input floats are three ds_read_b128 at a 48-byte lane stride (conflict-free per
//      16 lanes), its four values are ONE 16-byte global store (no LDS), its 12 gradient floats three ds_write_b128:
//      12 LDS instructions per tile instead of 35
// Any point count >= 256 and any 4-byte aligned buffers: the 16-byte accesses take dword addresses (common.h f32x4_u); a
// ragged end is covered by moving the last tile back so that it ends at the last point.
// Waves per workgroup.  Round 4: 16 (1024 threads, 64 KB of LDS, two workgroups per CU) instead of 8 -- found while A/B-ing
// LDS-DMA point loads (slower at every depth: profiles/r04_cq_variants.txt; the experiment's source is tools/patches/cq_ablate_and_dma.patch): the 64M-point launch 382 -> 346 us (5.43 TB/s =
// 0.68 of 8 TB/s; 336 us = 0.70 on another box), with EVERY point gathering 568 -> 419 us -- round 3's "gather ceiling that no
// launch geometry moves" (0.55-0.59 ms) was the 8-wave geometry's; 1M and 8M points unchanged (5.84 / 39.9 us).
#ifndef PVAMD_CQ_WAVES
#define PVAMD_CQ_WAVES 16
#endif
#ifndef PVAMD_CQ_OWN4
#define PVAMD_CQ_OWN4 1
#endif
constexpr int kWavesPerBlock = PVAMD_CQ_WAVES;
constexpr int kTilePoints = 256;
constexpr bool kOwn4 = PVAMD_CQ_OWN4 != 0;
#ifndef PVAMD_CQ_MIN_POINTS
#define PVAMD_CQ_MIN_POINTS 16384
#endif
constexpr int64_t kWaveTileMinPoints = PVAMD_CQ_MIN_POINTS;

// LD_NT / ST_NT: non-temporal loads / stores.  Measured (tools/kbench.hip): batches whose points are cache-resident
// (<= a few M points, e.g. produced by the previous kernel or re-used) prefer plain loads + nt stores (6.6 -> 6.3 us per
// 1M points); batches far beyond the 256 MB Infinity Cache prefer nt loads + plain stores (0.370 -> 0.358 ms per 64M).
#ifndef PVAMD_CQ_MINWAVES
#define PVAMD_CQ_MINWAVES 1
#endif
template <bool F64, bool WRITE_OOB, bool LD_NT, bool ST_NT>
__global__ __launch_bounds__(kWavesPerBlock * 64, PVAMD_CQ_MINWAVES) void cached_query_wave(const pvamd_grid_t g,
                                                                         const float* __restrict__ pts, int64_t P,
                                                                         float* __restrict__ val,
                                                                         float* __restrict__ grad,
                                                                         uint8_t* __restrict__ oob) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][1024];  // per wave: 768 floats xyz/grad + 256 floats val = 4 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* spf = lds[wave];
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(spf);
    float* svf = spf + 768;
    const int64_t ntiles = (P + kTilePoints - 1) / kTilePoints;
    const int64_t wstride = (int64_t)gridDim.x * kWavesPerBlock;
    int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    // any P >= 256: the LAST tile is moved back to end at the last point (it overlaps its neighbour; the overlap is
    // computed and written twice with the same bits), so every tile is whole and there is no partial-tile path
    auto first_point = [&](int64_t t) { return t * kTilePoints <= P - kTilePoints ? t * kTilePoints : P - kTilePoints; };
    f32x4 a, b, c;
    auto load3 = [&](int64_t t) {  // three contiguous KB
        const f32x4_u* src = reinterpret_cast<const f32x4_u*>(pts + 3 * first_point(t));
        if (LD_NT) {
            a = __builtin_nontemporal_load(src + lane);
            b = __builtin_nontemporal_load(src + lane + 64);
            c = __builtin_nontemporal_load(src + lane + 128);
        } else {
            a = src[lane];
            b = src[lane + 64];
            c = src[lane + 128];
        }
    };
    if (tile < ntiles) load3(tile);
    for (; tile < ntiles; tile += wstride) {
        sp[lane] = a;
        sp[lane + 64] = b;
        sp[lane + 128] = c;
        // software prefetch: the next tile's HBM loads are in flight while this tile is looked up (the wave fences
        // below stop the compiler from doing this itself); 0.47 -> 0.42 ms per 64M points (profiles/r01_kbench.txt)
        const int64_t next = tile + wstride;
        if (next < ntiles) load3(next);
        PVAMD_WAVE_SYNC();
        float px[4], py[4], pz[4];
        if constexpr (kOwn4) {
            const f32x4 q0 = sp[3 * lane], q1 = sp[3 * lane + 1], q2 = sp[3 * lane + 2];
            px[0] = q0.x; py[0] = q0.y; pz[0] = q0.z;
            px[1] = q0.w; py[1] = q1.x; pz[1] = q1.y;
            px[2] = q1.z; py[2] = q1.w; pz[2] = q2.x;
            px[3] = q2.y; py[3] = q2.z; pz[3] = q2.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = lane + 64 * k;
                px[k] = spf[3 * p];
                py[k] = spf[3 * p + 1];
                pz[k] = spf[3 * p + 2];
            }
        }
        PVAMD_WAVE_SYNC();
        const int64_t o = first_point(tile);
        f32x4 v4;
        if constexpr (kOwn4) {
            float4 r[4];
            bool valid[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = cached_lookup<F64, LD_NT>(g, px[k], py[k], pz[k], valid[k]);  // LD_NT = the streaming launch (> 8M points)
            }
            sp[3 * lane] = f32x4{r[0].y, r[0].z, r[0].w, r[1].y};
            sp[3 * lane + 1] = f32x4{r[1].z, r[1].w, r[2].y, r[2].z};
            sp[3 * lane + 2] = f32x4{r[2].w, r[3].y, r[3].z, r[3].w};
            v4 = f32x4{r[0].x, r[1].x, r[2].x, r[3].x};
            if constexpr (WRITE_OOB) {
                const uint32_t m = (valid[0] ? 0u : 1u) | (valid[1] ? 0u : 1u << 8) | (valid[2] ? 0u : 1u << 16) | (valid[3] ? 0u : 1u << 24);
                __builtin_memcpy(oob + o + 4 * lane, &m, 4);  // the 4 consecutive flags of this lane
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = lane + 64 * k;
                bool valid;
                const float4 r = cached_lookup<F64, LD_NT>(g, px[k], py[k], pz[k], valid);
                svf[p] = r.x;
                spf[3 * p] = r.y;
                spf[3 * p + 1] = r.z;
                spf[3 * p + 2] = r.w;
                if constexpr (WRITE_OOB) oob[o + p] = valid ? 0 : 1;
            }
        }
        PVAMD_WAVE_SYNC();
        f32x4_u* vdst = reinterpret_cast<f32x4_u*>(val + o);
        f32x4_u* dst = reinterpret_cast<f32x4_u*>(grad + 3 * o);
        if constexpr (!kOwn4) v4 = sp[192 + lane];
        if (ST_NT) {
            __builtin_nontemporal_store(v4, vdst + lane);
            __builtin_nontemporal_store(sp[lane], dst + lane);
            __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
            __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128);
        } else {
            vdst[lane] = v4;
            dst[lane] = sp[lane];
            dst[lane + 64] = sp[lane + 64];
            dst[lane + 128] = sp[lane + 128];
        }
        PVAMD_WAVE_SYNC();
    }
}

// ---- direct path (round 6): no LDS -- launches that fit the chip in about one round of resident waves (up to 8M points) ----
// Lane l of a wave owns points l, l + 64, ... (PPL of them) of the wave's tile of 64 x PPL consecutive points and moves each with
// 12-byte accesses at a 12-byte lane stride: every global_load_dwordx3 / global_store_dwordx3 covers a contiguous 768 B, the value
// store a contiguous 256 B -- the same bytes per instruction slot of the address path as the 16-byte accesses of the wave-tile
// kernel above, without the four LDS passes (12 KB per 256 points) and the wave fences that kernel's AoS <-> per-lane transposes
// need.  One tile per wave, all PPL point loads issued before the first look-up; non-temporal stores (with plain stores the
// 1M-point launch takes 6.9 us instead of 5.5; non-temporal LOADS cost 5-60 % while the points are cache-resident).  Against
// the wave-tile kernel, same box, tools/cq_sweep.py (profiles/r06_cq_direct.txt): 16K-512K points 4.5-5.0 -> 2.4-3.7 us, 1M points
// 5.56 -> 5.18 us (0.65 -> 0.71 of 8 TB/s), 4M 20.4 -> 18.9 us, 8M 36.9 -> 35.3 us (0.80 -> 0.83).  Beyond the Infinity Cache the
// wave-tile kernel stays: 64M points 337 us against 405-460 us in this form (its 16-byte accesses and its read-ahead matter there),
// and it keeps the sizes around 2M points where its 512 workgroups are exactly one round (9.13 against 9.3-9.5 us).
// Any point count >= 64 x PPL and any 4-byte aligned buffers; a ragged end moves the last tile back so that it ends at the last point.
template <bool F64, bool WRITE_OOB, int PPL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void cached_query_direct(const pvamd_grid_t g, const float* __restrict__ pts, int64_t P,
                                                                  float* __restrict__ val, float* __restrict__ grad,
                                                                  uint8_t* __restrict__ oob) {
    constexpr int kTile = 64 * PPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    if (tile * kTile >= P) return;  // wave-uniform
    const int64_t o = (tile * kTile <= P - kTile ? tile * kTile : P - kTile) + lane;
    float px[PPL], py[PPL], pz[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int64_t i = o + 64 * k;
        px[k] = pts[3 * i];
        py[k] = pts[3 * i + 1];
        pz[k] = pts[3 * i + 2];
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int64_t i = o + 64 * k;
        bool valid;
        const float4 r = cached_lookup<F64, false>(g, px[k], py[k], pz[k], valid);
        __builtin_nontemporal_store(r.x, val + i);
        __builtin_nontemporal_store(r.y, grad + 3 * i);
        __builtin_nontemporal_store(r.z, grad + 3 * i + 1);
        __builtin_nontemporal_store(r.w, grad + 3 * i + 2);
        if constexpr (WRITE_OOB) oob[i] = valid ? 0 : 1;
    }
}

// Which kernel serves P points (measured, profiles/r06_cq_direct.txt; one MI355X = 256 CUs x 32 resident waves):
//   < 16,384            one point per lane, grid-stride (cached_query_scalar)
//   .. 160K             direct, 1 point per lane, 8 waves      .. 896K   direct, 2 points per lane, 4 waves
//   .. 1M               direct, 2 points per lane, 16 waves (8192 waves in 512 workgroups: exactly two per CU)
//   .. 1.6M             direct, 4 points per lane, 4 waves      .. 2.25M  wave-tile kernel (512 workgroups = one round)
//   .. 8M               direct, 4 points per lane, 4 waves      beyond    wave-tile kernel, streaming instantiation
enum CqKind {
    kCqScalar = PVAMD_CQ_KERNEL_SCALAR, kCqDirect1 = PVAMD_CQ_KERNEL_DIRECT_1, kCqDirect2 = PVAMD_CQ_KERNEL_DIRECT_2,
    kCqDirect2Wide = PVAMD_CQ_KERNEL_DIRECT_2W, kCqDirect4 = PVAMD_CQ_KERNEL_DIRECT_4, kCqWaveTile = PVAMD_CQ_KERNEL_WAVE_TILE,
    kCqStreaming = PVAMD_CQ_KERNEL_STREAMING
};
static inline CqKind cq_kind(int64_t P) {
    if (P < kWaveTileMinPoints) return kCqScalar;
#ifdef PVAMD_CQ_NO_DIRECT  // A/B: the round-5 choice
    return P <= ((int64_t)8 << 20) ? kCqWaveTile : kCqStreaming;
#endif
    if (P <= 160 * 1024) return kCqDirect1;
    if (P <= 896 * 1024) return kCqDirect2;
    if (P <= 1024 * 1024) return kCqDirect2Wide;
    if (P <= 1600 * 1024) return kCqDirect4;
    if (P <= 2304 * 1024) return kCqWaveTile;
#ifndef PVAMD_CQ_STREAM_FROM
#define PVAMD_CQ_STREAM_FROM ((int64_t)8 << 20)
#endif
    if (P <= PVAMD_CQ_STREAM_FROM) return kCqDirect4;
    return kCqStreaming;
}

// ---- one point per lane: small batches (more waves than 256-point tiles would give) ----
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

// ---- float64 query points (sdf.py:545-547: output dtype = query dtype; torch promotion makes the index arithmetic,
// the range test and the BOUNDING_BOX branch float64).  One point per lane; 24 B read + 32 B written per point. ----
__global__ __launch_bounds__(256) void cached_query_f64_kernel(const pvamd_grid_t g, const double* __restrict__ pts,
                                                                int64_t P, double* __restrict__ val,
                                                                double* __restrict__ grad, uint8_t* __restrict__ oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        double v, gx, gy, gz;
        const bool valid = cached_lookup_f64(g, p, v, gx, gy, gz);
        val[i] = v;
        grad[3 * i] = gx;
        grad[3 * i + 1] = gy;
        grad[3 * i + 2] = gz;
        if (oob) oob[i] = valid ? 0 : 1;
    }
}

__global__ __launch_bounds__(256) void cached_outside_f64_kernel(const pvamd_grid_t g, const double* __restrict__ pts,
                                                                  int64_t P, double level, uint8_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        long long key[3];
        const bool valid = voxel_key_f64(g, p, key);
        // sdf.py:601: the float32 cache against a python scalar -- the scalar does not promote the tensor, so the comparison
        // is a float32 one whatever the dtype of the query points
        out[i] = valid ? (uint8_t)(g.vox[4 * (int64_t)clamped_flat(g, key)] > (float)level) : (uint8_t)1;
    }
}

__global__ __launch_bounds__(256) void voxel_index_f64_kernel(const pvamd_grid_t g, const double* __restrict__ pts,
                                                               int64_t P, int64_t* __restrict__ out_key,
                                                               int64_t* __restrict__ out_flat,
                                                               uint8_t* __restrict__ out_valid) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        long long key[3];
        const bool valid = voxel_key_f64(g, p, key);
        if (out_key) {
            out_key[3 * i] = key[0];
            out_key[3 * i + 1] = key[1];
            out_key[3 * i + 2] = key[2];
        }
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

extern "C" int pvamd_cached_query_kernel(int64_t P) { return (int)cq_kind(P); }

extern "C" int pvamd_cached_query(const pvamd_grid_t* grid, const float* points, int64_t P, float* out_val,
                                  float* out_grad, uint8_t* out_oob, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;  // empty query: nothing to read or write (torch hands out NULL for empty tensors)
    if (!grid || !out_val || !out_grad || !points) return PVAMD_E_NULL;
    if (int e = check_grid(*grid)) return e;
    if (!aligned_to(points, 4) || !aligned_to(out_val, 4) || !aligned_to(out_grad, 4)) return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const bool f64 = grid->index_f64 != 0;
    // One launch for any P (cq_kind above).
    const CqKind kind = cq_kind(P);
    if (kind == kCqWaveTile || kind == kCqStreaming) {
        const int64_t ntiles = (P + kTilePoints - 1) / kTilePoints;
        const int64_t need = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
// The streaming regime (> 8M points, far beyond the 256 MB Infinity Cache), round 3 (tools/cq_sweep.py over build
// variants, profiles/r03_cq_variants.txt; 64M points, 52 % out of range): one tile per wave instead of a capped grid with a
// grid-stride loop 431 -> 413 us, 8 waves per workgroup 369-374, non-temporal stores as well 361 us (5.2 TB/s); 16 waves
// 363; plain loads 382.  What bounds it is in profiles/r03_cq64_counters.md: the L1 (TCP) -> L2 request path, not HBM.
#ifndef PVAMD_CQ_BIG_ST_NT
#define PVAMD_CQ_BIG_ST_NT true
#endif
#ifndef PVAMD_CQ_BIG_LD_NT
#define PVAMD_CQ_BIG_LD_NT true
#endif
#ifndef PVAMD_CQ_BLOCKS
#define PVAMD_CQ_BLOCKS 1024
#endif
#ifndef PVAMD_CQ_BIG_BLOCKS
#define PVAMD_CQ_BIG_BLOCKS 0
#endif
        const bool big = kind == kCqStreaming;  // > 8M points (96 MB of xyz)
        const int64_t cap = big ? PVAMD_CQ_BIG_BLOCKS : PVAMD_CQ_BLOCKS;  // 0: one tile per wave, no grid-stride loop
        const dim3 grid_dim((unsigned)((cap > 0 && need > cap) ? cap : need)), block(kWavesPerBlock * 64);
#define PVAMD_LAUNCH_CQ(F64_, OOB_)                                                                                      \
    do {                                                                                                                \
        if (big) hipLaunchKernelGGL((cached_query_wave<F64_, OOB_, PVAMD_CQ_BIG_LD_NT, PVAMD_CQ_BIG_ST_NT>), grid_dim, block, 0, s, *grid, points, P, out_val, out_grad, out_oob); \
        else hipLaunchKernelGGL((cached_query_wave<F64_, OOB_, false, true>), grid_dim, block, 0, s, *grid, points, P, out_val, out_grad, out_oob);    \
    } while (0)
        if (f64) {
            if (out_oob) PVAMD_LAUNCH_CQ(true, true);
            else PVAMD_LAUNCH_CQ(true, false);
        } else {
            if (out_oob) PVAMD_LAUNCH_CQ(false, true);
            else PVAMD_LAUNCH_CQ(false, false);
        }
#undef PVAMD_LAUNCH_CQ
    } else if (kind != kCqScalar) {
#define PVAMD_LAUNCH_DIRECT(PPL_, WAVES_)                                                                               \
    do {                                                                                                                \
        const int64_t tiles_ = (P + 64 * PPL_ - 1) / (64 * PPL_);                                                       \
        const dim3 grid_dim((unsigned)((tiles_ + WAVES_ - 1) / WAVES_)), block(WAVES_ * 64);                            \
        if (f64) {                                                                                                      \
            if (out_oob) hipLaunchKernelGGL((cached_query_direct<true, true, PPL_, WAVES_>), grid_dim, block, 0, s, *grid, points, P, out_val, out_grad, out_oob);   \
            else hipLaunchKernelGGL((cached_query_direct<true, false, PPL_, WAVES_>), grid_dim, block, 0, s, *grid, points, P, out_val, out_grad, out_oob);        \
        } else {                                                                                                        \
            if (out_oob) hipLaunchKernelGGL((cached_query_direct<false, true, PPL_, WAVES_>), grid_dim, block, 0, s, *grid, points, P, out_val, out_grad, out_oob);  \
            else hipLaunchKernelGGL((cached_query_direct<false, false, PPL_, WAVES_>), grid_dim, block, 0, s, *grid, points, P, out_val, out_grad, out_oob);       \
        }                                                                                                               \
    } while (0)
        if (kind == kCqDirect1) PVAMD_LAUNCH_DIRECT(1, 8);
        else if (kind == kCqDirect2) PVAMD_LAUNCH_DIRECT(2, 4);
        else if (kind == kCqDirect2Wide) PVAMD_LAUNCH_DIRECT(2, 16);
        else PVAMD_LAUNCH_DIRECT(4, 4);
#undef PVAMD_LAUNCH_DIRECT
    } else {
        const dim3 grid_dim(stream_grid(P, 256)), block(256);
        if (f64) hipLaunchKernelGGL((cached_query_scalar<true>), grid_dim, block, 0, s, *grid, points, (int64_t)0, P, out_val, out_grad, out_oob);
        else hipLaunchKernelGGL((cached_query_scalar<false>), grid_dim, block, 0, s, *grid, points, (int64_t)0, P, out_val, out_grad, out_oob);
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

extern "C" int pvamd_cached_query_f64(const pvamd_grid_t* grid, const double* points, int64_t P, double* out_val,
                                      double* out_grad, uint8_t* out_oob, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grid || !out_val || !out_grad || !points) return PVAMD_E_NULL;
    if (int e = check_grid(*grid)) return e;
    if (!aligned_to(points, 8) || !aligned_to(out_val, 8) || !aligned_to(out_grad, 8)) return PVAMD_E_ALIGN;
    hipLaunchKernelGGL(cached_query_f64_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, (hipStream_t)stream, *grid, points,
                       P, out_val, out_grad, out_oob);
    return (int)hipGetLastError();
}

extern "C" int pvamd_cached_outside_f64(const pvamd_grid_t* grid, const double* points, int64_t P, double level,
                                        uint8_t* out, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grid || !out || !points) return PVAMD_E_NULL;
    if (int e = check_grid(*grid)) return e;
    if (!aligned_to(points, 8)) return PVAMD_E_ALIGN;
    hipLaunchKernelGGL(cached_outside_f64_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, (hipStream_t)stream, *grid,
                       points, P, level, out);
    return (int)hipGetLastError();
}

extern "C" int pvamd_voxel_index_f64(const pvamd_grid_t* grid, const double* points, int64_t P, int64_t* out_key,
                                     int64_t* out_flat, uint8_t* out_valid, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grid || !points) return PVAMD_E_NULL;
    if (int e = check_grid(*grid, /*need_vox=*/false)) return e;
    if (!aligned_to(points, 8)) return PVAMD_E_ALIGN;
    hipLaunchKernelGGL(voxel_index_f64_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, (hipStream_t)stream, *grid, points,
                       P, out_key, out_flat, out_valid);
    return (int)hipGetLastError();
}
