// The one-workgroup spatial sort of up to 16384 points (1024 threads).  gfx950 only.
#pragma once
#include "common.h"
#include "morton.h"

namespace pvamd {

// ---- up to 16384 points: the whole thing in ONE workgroup (bounds, 16^3-cell histogram in LDS, scan, scatter) ----
// The seven launches above cost ~4.5 us each whatever their size; for the 10k-point query of BASELINE C1 that was a
// quarter of the call.  Every thread keeps its (up to 16) points and their cells in registers: the points are read once,
// the curve position is worked out once, and the scan of the 4096 cell counters is a wave scan + 16 wave totals (five
// barriers in all; the first version read the points three times and scanned with twenty barriers: 25 us for 10k points).
constexpr int kSmallCells = 4096, kSmallPer = 16;  // 16^3 cells: the leading 12 bits of a 30-bit key
// (a device function so that the few-points mesh query can run it in ONE block of a launch whose other blocks do the work
// that does not need the order: csrc/mesh.hip, mesh_small_prep_kernel)
PVAMD_DEV void order_small_block(const float* __restrict__ pts, int P, int* __restrict__ order,
                                 int* __restrict__ inv, float* __restrict__ sorted_pts) {
    __shared__ unsigned hist[kSmallCells];
    __shared__ unsigned wsum[16];
    __shared__ float part[16][6];
    __shared__ float box[6];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float x[kSmallPer], y[kSmallPer], z[kSmallPer];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < kSmallPer; ++k) {
        const int i = t + 1024 * k;
        x[k] = y[k] = z[k] = NAN;
        if (i < P) { x[k] = pts[3 * i]; y[k] = pts[3 * i + 1]; z[k] = pts[3 * i + 2]; }
        const float v[3] = {x[k], y[k], z[k]};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (fabsf(v[d]) < INFINITY) {  // false for NaN and +-inf
                lo[d] = fminf(lo[d], v[d]);
                hi[d] = fmaxf(hi[d], v[d]);
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
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { part[wave][d] = lo[d]; part[wave][3 + d] = hi[d]; }
    }
#pragma unroll
    for (int k = 0; k < kSmallCells / 1024; ++k) hist[t + 1024 * k] = 0u;
    __syncthreads();
    if (t < 6) {
        float v = part[0][t];
        for (int w = 1; w < 16; ++w) v = t < 3 ? fminf(v, part[w][t]) : fmaxf(v, part[w][t]);
        box[t] = v;
    }
    __syncthreads();
    const float blo[3] = {box[0], box[1], box[2]}, bhi[3] = {box[3], box[4], box[5]};
    float scale[3];  // as hilbert_key30 with b = 4
#pragma unroll
    for (int d = 0; d < 3; ++d) scale[d] = (15.f + 0.999f) / fmaxf(bhi[d] - blo[d], 1e-30f);
    unsigned cell[kSmallPer];
#pragma unroll
    for (int k = 0; k < kSmallPer; ++k) {
        cell[k] = 0u;
        if (1024 * k < P) {  // uniform over the block: the one workgroup is bound by its vector ALUs
#ifdef PVAMD_ORDER_MORTON
            cell[k] = morton_key30(x[k], y[k], z[k], blo, bhi) >> 18;
#else
            cell[k] = hilbert_cell16(x[k], y[k], z[k], blo, scale);
#endif
            if (t + 1024 * k < P) atomicAdd(&hist[cell[k]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the 4096 counters: thread t owns cells 4t .. 4t+3
    unsigned c4[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c4[k] = hist[4 * t + k]; sum += c4[k]; }
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) { hist[4 * t + k] = run; run += c4[k]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSmallPer; ++k) {
        const int i = t + 1024 * k;
        if (i < P) {
            const unsigned slot = atomicAdd(&hist[cell[k]], 1u);
            order[slot] = i;
            if (inv) inv[i] = (int)slot;
            if (sorted_pts) {
                sorted_pts[3 * slot] = x[k];
                sorted_pts[3 * slot + 1] = y[k];
                sorted_pts[3 * slot + 2] = z[k];
            }
        }
    }
}

}  // namespace pvamd
