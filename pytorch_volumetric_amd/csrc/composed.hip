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
    bool unnormalised;    // (gx,gy,gz) is the bounding-box vector t; the gradient t/v is formed only for the winner
};

template <bool ANY_F64>
PVAMD_DEV void visit_leaf(const pvamd_grid_t& g, const float* __restrict__ M, int s, float px, float py, float pz,
                          Best& best) {
    const float x = affine_row(M[0], M[1], M[2], M[3], px, py, pz);
    const float y = affine_row(M[4], M[5], M[6], M[7], px, py, pz);
    const float z = affine_row(M[8], M[9], M[10], M[11], px, py, pz);
    float v, a, b, c;
    const bool valid = in_range(g, x, y, z);
    if (valid) {
        const int flat = (ANY_F64 && g.index_f64) ? voxel_flat_in_range<true>(g, x, y, z)
                                                  : voxel_flat_in_range<false>(g, x, y, z);
        const float4 r = reinterpret_cast<const float4*>(g.vox)[flat];
        v = r.x; a = r.y; b = r.z; c = r.w;
    } else {
        float t[3];
        v = bounding_box_vector(g, x, y, z, t);  // out-of-range leaves cost no division unless they win
        a = t[0]; b = t[1]; c = t[2];
    }
    // torch.argmin semantics (sdf.py:421): first minimum wins, NaN counts as the minimum
    const bool take = (best.s < 0) || (v < best.v) || (v != v && best.v == best.v);
    if (take) {
        best.v = v;
        best.gx = a;
        best.gy = b;
        best.gz = c;
        best.s = s;
        best.unnormalised = !valid;
    }
}

// g_obj = R^T g_leaf with R the obj->leaf rotation (sdf.py:409 transform_normals by the inverse transform)
PVAMD_DEV void rotate_back(const float* __restrict__ M, const Best& b, float& ox, float& oy, float& oz) {
    float gx = b.gx, gy = b.gy, gz = b.gz;
    if (b.unnormalised) {  // sdf.py:570 grad = dtotal / dist
        gx = div_rn(gx, b.v);
        gy = div_rn(gy, b.v);
        gz = div_rn(gz, b.v);
    }
    ox = fmaf(M[8], gz, fmaf(M[4], gy, mul_rn(M[0], gx)));
    oy = fmaf(M[9], gz, fmaf(M[5], gy, mul_rn(M[1], gx)));
    oz = fmaf(M[10], gz, fmaf(M[6], gy, mul_rn(M[2], gx)));
}

// ---- leaf culling ----
// Per (leaf, configuration) a sphere in the OBJECT frame: centre = centre of the leaf's valid range box, squared
// radius of that box (a point farther than that is certainly out of range, so its value is the bounding-box
// distance), and the radius of the leaf's surface bounding box about the same centre (that distance is at least
// |p - c| - r_bb).  A leaf whose lower bound cannot beat the running minimum of any of the wave's 256 points is skipped
// as a whole; the bounds are inflated, so skipping never changes a result (first-minimum semantics need a strictly
// smaller value to replace the incumbent).  Pays off for spatially coherent queries (grids, slices, scans).
constexpr int kMaxCullLeaves = 64;

PVAMD_DEV void build_cull_spheres(const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A,
                                  int a, float (*cull)[8]) {
    for (int s = threadIdx.x; s < S && s < kMaxCullLeaves; s += blockDim.x) {
        const pvamd_grid_t& g = grids[s];
        const float* M = tf + 16 * ((int64_t)s * A + a);
        float cl[3], r2 = 0.f, e2 = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cl[d] = 0.5f * (g.vlo[d] + g.vhi[d]);
            const float h = 0.5f * (g.vhi[d] - g.vlo[d]);
            r2 += h * h;
            const float e = fmaxf(fabsf(g.bb_min[d] - cl[d]), fabsf(g.bb_max[d] - cl[d]));
            e2 += e * e;
        }
        // c_obj = R^T (c_leaf - t)
        const float ux = cl[0] - M[3], uy = cl[1] - M[7], uz = cl[2] - M[11];
        cull[s][0] = M[0] * ux + M[4] * uy + M[8] * uz;
        cull[s][1] = M[1] * ux + M[5] * uy + M[9] * uz;
        cull[s][2] = M[2] * ux + M[6] * uy + M[10] * uz;
        const float scale = fmaxf(fmaxf(fabsf(cl[0]), fabsf(cl[1])), fabsf(cl[2])) + sqrt_rn(r2) + sqrt_rn(e2);
        const float rr = sqrt_rn(r2) * 1.0001f + 1e-5f * scale;
        cull[s][3] = rr * rr;                                   // beyond this (squared) the point is out of range
        cull[s][4] = sqrt_rn(e2) * 1.0001f + 1e-5f * scale;    // radius of the surface bounding box
        cull[s][5] = cull[s][6] = cull[s][7] = 0.f;
    }
}

// 1 when leaf `c` provably cannot replace the incumbent minimum of this point.  Straight-line (no short-circuit): the
// compiler turned the && / || form into a tree of exec-mask branches that kept the scalar unit busier than the test.
PVAMD_DEV int leaf_cannot_win(const float* __restrict__ c, float px, float py, float pz, const Best& best) {
    const float dx = px - c[0], dy = py - c[1], dz = pz - c[2];
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float t = best.v + c[4];
    const int has_best = best.s >= 0;
    const int out_of_range = d2 > c[3];
    const int beaten = (int)(t <= 0.f) | (int)(d2 >= t * t * 1.0003f);
    return has_best & out_of_range & beaten;
}

// One wave = 256 consecutive points of one configuration per pass; all global traffic in contiguous 1 KB pieces
// through a wave-private LDS slice (same scheme as cached_query_wave, see cached.hip).
constexpr int kWavesPerBlock = 4;
constexpr int kTilePoints = 256;
#ifndef PVAMD_COMPOSED_PPP
#define PVAMD_COMPOSED_PPP 2
#endif

template <bool ANY_F64, int PPP>
__global__ __launch_bounds__(kWavesPerBlock * 64) void composed_query_wave(const pvamd_grid_t* __restrict__ grids, int S,
                                                                           const float* __restrict__ tf, int A,
                                                                           const f32x4* __restrict__ pts4,
                                                                           int64_t ntiles, int64_t P,
                                                                           float* __restrict__ val,
                                                                           float* __restrict__ grad,
                                                                           int* __restrict__ leaf, int a0) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][1024];
    __shared__ float cull[kMaxCullLeaves][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* spf = lds[wave];
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(spf);
    float* svf = spf + 768;
    const int a = a0 + blockIdx.y;
    build_cull_spheres(grids, S, tf, A, a, cull);
    __syncthreads();
    const int64_t wstride = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += wstride) {
        const f32x4* src = pts4 + tile * 192;  // re-read for every configuration: L2-resident
        sp[lane] = src[lane];
        sp[lane + 64] = src[lane + 64];
        sp[lane + 128] = src[lane + 128];
        PVAMD_WAVE_SYNC();
        // PPP points per lane go through the leaf loop together (fewer live registers -> more waves per SIMD; the
        // leaf constants are scalar loads, so re-walking the leaves per pass costs SALU/SMEM, not VALU)
#pragma unroll
        for (int h = 0; h < 4; h += PPP) {
            float px[PPP], py[PPP], pz[PPP];
            Best best[PPP];
#pragma unroll
            for (int k = 0; k < PPP; ++k) {
                const int p = lane + 64 * (h + k);
                px[k] = spf[3 * p];
                py[k] = spf[3 * p + 1];
                pz[k] = spf[3 * p + 2];
                best[k] = Best{0.f, 0.f, 0.f, 0.f, -1, false};
            }
            for (int s = 0; s < S; ++s) {
                if (s < kMaxCullLeaves) {
                    const float* c = cull[s];  // wave-uniform: LDS broadcast
                    int dead = 1;
#pragma unroll
                    for (int k = 0; k < PPP; ++k) dead &= leaf_cannot_win(c, px[k], py[k], pz[k], best[k]);
                    if (__all(dead)) continue;
                }
                const float* M = tf + 16 * ((int64_t)s * A + a);  // wave-uniform: scalar loads
                const pvamd_grid_t& g = grids[s];
#pragma unroll
                for (int k = 0; k < PPP; ++k) visit_leaf<ANY_F64>(g, M, s, px[k], py[k], pz[k], best[k]);
            }
#pragma unroll
            for (int k = 0; k < PPP; ++k) {
                const int p = lane + 64 * (h + k);
                // per-lane winner: these matrix reads are vector loads, but the S*A stack is tiny and cache-resident
                const float* M = tf + 16 * ((int64_t)best[k].s * A + a);
                float gx, gy, gz;
                rotate_back(M, best[k], gx, gy, gz);
                svf[p] = best[k].v;  // a lane overwrites only the LDS slots of the points it owns
                spf[3 * p] = gx;
                spf[3 * p + 1] = gy;
                spf[3 * p + 2] = gz;
                if (leaf) leaf[(int64_t)a * P + tile * kTilePoints + p] = best[k].s;
            }
        }
        PVAMD_WAVE_SYNC();
        const int64_t o = (int64_t)a * P + tile * kTilePoints;  // multiple of 4: rows start 16-byte aligned
        __builtin_nontemporal_store(sp[192 + lane], reinterpret_cast<f32x4*>(val + o) + lane);
        f32x4* dst = reinterpret_cast<f32x4*>(grad + 3 * o);
        __builtin_nontemporal_store(sp[lane], dst + lane);
        __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
        __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128);
        PVAMD_WAVE_SYNC();
    }
}

template <bool ANY_F64>
__global__ __launch_bounds__(256) void composed_query_scalar(const pvamd_grid_t* __restrict__ grids, int S,
                                                              const float* __restrict__ tf, int A,
                                                              const float* __restrict__ pts, int64_t first,
                                                              int64_t P, float* __restrict__ val,
                                                              float* __restrict__ grad, int* __restrict__ leaf, int a0) {
    const int a = a0 + blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        Best best{0.f, 0.f, 0.f, 0.f, -1, false};
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
    if (S < 1 || A < 1 || P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grids || !tf || !out_val || !out_grad || !points) return PVAMD_E_NULL;
    if (!aligned_to(grids, 8) || !aligned_to(tf, 4) || !aligned_to(points, 4)) return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const bool vec_ok = (P % 4 == 0) && aligned_to(points, 16) && aligned_to(out_val, 16) && aligned_to(out_grad, 16);
    // The leaf descriptors live in device memory; whether any of them asks for float64 index arithmetic is not
    // known host-side, so the kernels are built for the general case and test the (wave-uniform) flag per leaf.
    // The wave-tile kernel walks 4 points per lane one after the other: it wins once there are enough 256-point tiles to
    // fill the chip (1024 SIMDs x a few waves); below that the one-point-per-lane kernel has 4x the parallelism and a
    // quarter of the latency (100k points x 8 leaves: 36 -> see profiles/ latency numbers).
    const bool enough = vec_ok && (P / kTilePoints) * (int64_t)A >= 4096;
    const int64_t ntiles = enough ? P / kTilePoints : 0;
    // up to ~65536 blocks in total, split over the A configurations: about one 256-point tile per wave.  (Sweep on C4,
    // 200 x 262,144: 1024 blocks 1.40 ms, 4096 1.17, 8192 1.13, 32768 1.09, 65536 1.08 -- the hardware dispatcher
    // balances better than a grid-stride loop over unequal tiles.)
    // gridDim.y carries the configuration: at most 65535 per launch, so larger batches go out in slabs (the kernels
    // take the slab's first configuration and index transforms / outputs with the global one)
    constexpr int kSlab = 65535;
    for (int a0 = 0; a0 < A; a0 += kSlab) {
        const int An = A - a0 < kSlab ? A - a0 : kSlab;
        const int64_t cap = ((int64_t)65536 + An - 1) / An;
        if (ntiles > 0) {
            const int64_t need = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
            const unsigned gx = (unsigned)(need < cap ? need : (cap < 1 ? 1 : cap));
            hipLaunchKernelGGL((composed_query_wave<true, PVAMD_COMPOSED_PPP>), dim3(gx, An), dim3(kWavesPerBlock * 64), 0, s, grids, S, tf, A,
                               reinterpret_cast<const f32x4*>(points), ntiles, P, out_val, out_grad, out_leaf, a0);
        }
        const int64_t first = ntiles * kTilePoints;
        if (first < P) {
            const int64_t need = (P - first + 255) / 256;
            const unsigned gx = (unsigned)(need < cap ? need : (cap < 1 ? 1 : cap));
            hipLaunchKernelGGL((composed_query_scalar<true>), dim3(gx, An), dim3(256), 0, s, grids, S, tf, A, points, first,
                               P, out_val, out_grad, out_leaf, a0);
        }
    }
    return (int)hipGetLastError();
}
