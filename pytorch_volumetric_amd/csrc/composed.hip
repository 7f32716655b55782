// ComposedSDF / RobotSDF query kernel (BASELINE configs C3, C4): for every (configuration a, point p) walk the S
// leaves in registers -- 3x4 affine into the leaf frame, nearest-voxel gather of the packed (val, grad) record (or the
// bounding-box fallback), running first-minimum -- rotate the winning gradient back once, and write 16 B.
// Replaces sdf.py:392-433 of the reference (transform_points broadcast to (S*A, P, 3), a Python loop of S
// CachedSDF calls, cat, argmin, gather): none of those intermediates reaches HBM here.
//
// Leaf descriptors and the S transforms of configuration a are wave-uniform, so they are read through the scalar
// cache into SGPRs (no LDS round trip); blockIdx.y = a keeps them uniform for the whole block.
//
// The kernel is VALU-issue-bound (profiles/r02_*: SQ_ACTIVE_INST_VALU = the kernel's duration), so this file counts
// vector instructions:
//  * the bounding-box vector of sdf.py:559-567 is ONE v_med3_f32 per component: with d1 = x - bb_min >= d2 = x - bb_max,
//    "negate (bb_min - x) where positive, else max(x - bb_max, 0)" is the median of (d1, d2, 0), exactly (negation and
//    adding 0 are exact, x < bb_min <=> bb_min - x > 0);
//  * its norm is the correctly rounded square root built from v_sqrt_f32 + the two-sided residual test (what the
//    compiler emits, minus the denormal pre-scaling, which a one-compare guard sends to the generic path);
//  * the out-of-range candidate is computed branch-free for all lanes and the gather only under the in-range mask;
//  * whole leaves are skipped per 256-point wave tile from ONE bounding-sphere test evaluated by 64 lanes = 64 leaves
//    in parallel (instead of a per-lane test in front of every visit).
#include "common.h"
#include "grid_lookup.h"
#include "wave_ops.h"
#include "morton.h"

namespace pvamd {

constexpr int kUnnormalised = 1 << 30;
struct Best {
    float v, gx, gy, gz;  // gradient kept in the winning leaf's frame until the end
    int tag;              // winning leaf | kUnnormalised when (gx,gy,gz) is the bounding-box vector t (gradient = t / v).
                          // (Tracking that bit as a wave mask in SGPRs instead saves one v_cndmask per visit and costs
                          // five scalar instructions: measured slower, C4 0.92 -> 0.94 ms -- the scalar unit is as busy
                          // as the vector units in this kernel.)
};

// How a visit obtains its voxel index (all three give the reference's index, bit for bit):
//   kEstimate    the fp32 estimate; `unsure` is raised where it cannot be trusted and the CALLER redoes those points with
//                kExact after its leaf loop -- keeps the division sequence and its registers out of the hot loop; right
//                when flags are rare (a few per million visits on 100 KB link grids)
//   kInlineExact the estimate with the exact statements inline behind one rare branch -- right when flags are common
//                (large coordinate / resolution ratios: 21 MB README-size link grids flag 22 % of the wave passes, and
//                a redo pass per flag costs 6.2 -> 7.1 ms there)
//   kExact       the reference's own statements only (IEEE division in the leaf's index dtype)
enum IndexMode { kEstimate = 0, kInlineExact = 1, kExact = 2 };

// One leaf for one point: candidate (v, a, b, c) and whether it came from the grid (valid) or is the unnormalised
// bounding-box vector.
// The range test leaves as a LANE MASK (an SGPR pair), not as a bool: a bool that crosses basic blocks -- let alone one kept
// in a `bool valid[PPP]` array -- is materialised as 0 / 1 in a VGPR (v_cndmask) and compared against 0 again at every use
// (v_cmp_ne); the mask is tested wave-wide with scalar compares and turned back into a per-lane predicate for free
// (inverse_ballot: the v_cndmask that consumes it reads the SGPR pair directly).
template <int MODE>
PVAMD_DEV void leaf_candidate(const pvamd_grid_t& g, const float* __restrict__ M, float px, float py, float pz,
                               float& v, float& a, float& b, float& c, uint64_t& vmask, bool& unsure) {
    const float x = affine_row(M[0], M[1], M[2], M[3], px, py, pz);
    const float y = affine_row(M[4], M[5], M[6], M[7], px, py, pz);
    const float z = affine_row(M[8], M[9], M[10], M[11], px, py, pz);
    vmask = in_range_mask(g, x, y, z);
    const bool valid = __builtin_amdgcn_inverse_ballot_w64(vmask);
    auto gather = [&]() {
        int flat;
        if constexpr (MODE == kExact) {
            if (g.index_f64) voxel_flat<true>(g, x, y, z, flat);
            else voxel_flat<false>(g, x, y, z, flat);
        } else if constexpr (MODE == kInlineExact) {
            flat = voxel_flat_in_range_fused(g, x, y, z);
        } else {
            flat = voxel_flat_estimate(g, x, y, z, unsure);
        }
        const float4 r = load_record(g.vox, flat);
        v = r.x; a = r.y; b = r.z; c = r.w;
    };
    if (vmask == __builtin_amdgcn_ballot_w64(true)) {  // wave-uniform: the whole wave is inside this leaf's range
        gather();
        return;
    }
    // sdf.py:559-567 for every lane (no exec masking; the in-range lanes' results are overwritten below)
    a = __builtin_amdgcn_fmed3f(sub_rn(x, g.bb_min[0]), sub_rn(x, g.bb_max[0]), 0.f);
    b = __builtin_amdgcn_fmed3f(sub_rn(y, g.bb_min[1]), sub_rn(y, g.bb_max[1]), 0.f);
    c = __builtin_amdgcn_fmed3f(sub_rn(z, g.bb_min[2]), sub_rn(z, g.bb_max[2]), 0.f);
    v = sqrt_rn_sumsq(fmaf(c, c, fmaf(b, b, mul_rn(a, a))));  // sdf.py:568; the division of :570 waits for the winner
    if (vmask != 0) {
        if (valid) gather();
    }
}

// All leaves of the mask for one point, first-minimum semantics (see keep_first_minimum).
template <int MODE>
PVAMD_DEV void walk_leaves(const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A, int a,
                            uint64_t todo, float px, float py, float pz, struct Best& best, bool& unsure);

// torch.argmin semantics (sdf.py:421): first minimum wins, NaN counts as the minimum.  `best` starts at +inf, so
// "v is smaller, or v is NaN and the incumbent is not" is !(v >= best.v) && best.v == best.v.
PVAMD_DEV void keep_first_minimum(Best& best, int s, float v, float a, float b, float c, bool valid) {
    const bool take = !(v >= best.v) & (best.v == best.v);
    best.v = take ? v : best.v;
    best.gx = take ? a : best.gx;
    best.gy = take ? b : best.gy;
    best.gz = take ? c : best.gz;
    best.tag = take ? (valid ? s : (s | kUnnormalised)) : best.tag;
}

// The state before any leaf: +inf loses to every finite value and to NaN; if every leaf answers +inf (an infinite
// query coordinate) the first visited leaf stays the winner with the gradient inf/inf = NaN the reference gets.
PVAMD_DEV Best best_init(int first_leaf) {
    return Best{__builtin_inff(), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), first_leaf | kUnnormalised};
}

template <int MODE>
PVAMD_DEV void walk_leaves(const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A, int a,
                            uint64_t todo, float px, float py, float pz, Best& best, bool& unsure) {
    for (int s = 0; s < S; ++s) {
        if (s < 64 && !((todo >> s) & 1ull)) continue;  // wave-uniform
        float v, ga, gb, gc;
        uint64_t vm;
        leaf_candidate<MODE>(grids[s], tf + 16 * ((int64_t)s * A + a), px, py, pz, v, ga, gb, gc, vm, unsure);
        keep_first_minimum(best, s, v, ga, gb, gc, __builtin_amdgcn_inverse_ballot_w64(vm));
    }
}

// g_obj = R^T g_leaf with R the obj->leaf rotation (sdf.py:409 transform_normals by the inverse transform)
PVAMD_DEV void rotate_back(const float* __restrict__ M, const Best& b, float& ox, float& oy, float& oz) {
    float gx = b.gx, gy = b.gy, gz = b.gz;
    if (b.tag & kUnnormalised) {  // sdf.py:570 grad = dtotal / dist
        gx = div_rn(gx, b.v);
        gy = div_rn(gy, b.v);
        gz = div_rn(gz, b.v);
    }
    ox = fmaf(M[8], gz, fmaf(M[4], gy, mul_rn(M[0], gx)));
    oy = fmaf(M[9], gz, fmaf(M[5], gy, mul_rn(M[1], gx)));
    oz = fmaf(M[10], gz, fmaf(M[6], gy, mul_rn(M[2], gx)));
}

// ---- leaf culling ----
// Per (leaf, configuration), in the OBJECT frame (rigid transforms preserve distances): c = centre of the leaf's valid
// range box; r_range = radius of that box about c (a point farther away is certainly out of range, so its value is the
// bounding-box distance); r_bb = radius of the surface bounding box about c (that distance is >= |p - c| - r_bb);
// e = distance from c to the bounding box (that distance is <= |p - c| + e).  All inflated, so that skipping a leaf
// never changes a result: a leaf is skipped for a 256-point tile only when every point of the tile is out of its range
// AND its lower bound exceeds the upper bound some other out-of-range leaf guarantees for every point (strictly, with
// margin -- first-minimum ties cannot be affected).
constexpr int kMaxCullLeaves = 64;
#ifndef PVAMD_COMPOSED_MASK_SPAN
#define PVAMD_COMPOSED_MASK_SPAN 0.5f
#endif
constexpr float kMaskSpan = PVAMD_COMPOSED_MASK_SPAN;

PVAMD_DEV void build_cull_spheres(const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A,
                                  int a, float (*cull)[8]) {
    for (int s = threadIdx.x; s < S && s < kMaxCullLeaves; s += blockDim.x) {
        const pvamd_grid_t& g = grids[s];
        const float* M = tf + 16 * ((int64_t)s * A + a);
        float cl[3], r2 = 0.f, e2 = 0.f, q2 = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cl[d] = 0.5f * (g.vlo[d] + g.vhi[d]);
            const float h = 0.5f * (g.vhi[d] - g.vlo[d]);
            r2 += h * h;
            const float e = fmaxf(fabsf(g.bb_min[d] - cl[d]), fabsf(g.bb_max[d] - cl[d]));
            e2 += e * e;
            const float q = fmaxf(fmaxf(g.bb_min[d] - cl[d], cl[d] - g.bb_max[d]), 0.f);  // c to the box, per axis
            q2 += q * q;
        }
        // c_obj = R^T (c_leaf - t)
        const float ux = cl[0] - M[3], uy = cl[1] - M[7], uz = cl[2] - M[11];
        cull[s][0] = M[0] * ux + M[4] * uy + M[8] * uz;
        cull[s][1] = M[1] * ux + M[5] * uy + M[9] * uz;
        cull[s][2] = M[2] * ux + M[6] * uy + M[10] * uz;
        const float scale = fmaxf(fmaxf(fabsf(cl[0]), fabsf(cl[1])), fabsf(cl[2])) + sqrt_rn(r2) + sqrt_rn(e2) +
                            fabsf(M[3]) + fabsf(M[7]) + fabsf(M[11]);
        cull[s][3] = sqrt_rn(r2) * 1.0001f + 1e-5f * scale;  // r_range
        cull[s][4] = sqrt_rn(e2) * 1.0001f + 1e-5f * scale;  // r_bb
        cull[s][5] = sqrt_rn(q2) * 1.0001f + 1e-5f * scale;  // e
        cull[s][6] = cull[s][7] = 0.f;
    }
}

// The leaves a set of points inside the sphere (ct, rt) may need (all coordinates at most `mag` in magnitude); lane = leaf.
PVAMD_DEV uint64_t leaf_mask_of_sphere(const float (*cull)[8], int S, int lane, const float ct[3], float rt, float mag,
                                       float& lower) {
    bool near = false;
    float upper = __builtin_inff();
    lower = -__builtin_inff();
    if (lane < S) {
        const float* c = cull[lane];  // S <= 64 rows of 8 floats: lanes read distinct rows
        const float dx = ct[0] - c[0], dy = ct[1] - c[1], dz = ct[2] - c[2];
        const float d = fast_sqrt(dx * dx + dy * dy + dz * dz);
        const float slack = 1e-5f * (d + mag) + 1e-30f;
        near = !(d * 0.9999f - slack > rt + c[3]);           // some point may be inside the leaf's range
        // when far: every point's value for this leaf is >= lower and <= upper
        lower = near ? -__builtin_inff() : d * 0.9999f - slack - rt - c[4];
        upper = near ? __builtin_inff() : d * 1.0001f + slack + rt + c[5];
    }
    const float ub = wave_min(upper);
    const bool visit = !(lower > ub);
    return __builtin_amdgcn_ballot_w64(visit && lane < S) | (S > 64 ? ~0ull : 0ull);
}

// Leaves (bit s of the result) that some point of the wave's 256-point tile may need.  lane = leaf.  `lower` (lane s)
// = a lower bound of leaf s's value over the whole tile when the tile is entirely outside the leaf's range, -inf
// otherwise: what the leaf loop re-tests against its running minimum (see composed_query_wave).
PVAMD_DEV uint64_t tile_leaf_mask(const float (*cull)[8], int S, int lane, const float* __restrict__ spf, float& lower) {
    float lo[3], hi[3];
    bool odd = false;  // a NaN / infinite coordinate: no bounds, visit everything
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = __builtin_inff();
        hi[d] = -__builtin_inff();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = lane + 64 * k;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float x = spf[3 * p + d];
            lo[d] = fminf(lo[d], x);
            hi[d] = fmaxf(hi[d], x);
            odd |= !(fabsf(x) < __builtin_inff());
        }
    }
    const uint64_t all = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
    lower = -__builtin_inff();
    if (wave_any(odd)) return all;
    float ct[3], rt2 = 0.f, mag = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float l = wave_min(lo[d]), h = wave_max(hi[d]);
        ct[d] = 0.5f * (l + h);
        const float half = 0.5f * (h - l);
        rt2 += half * half;
        mag = fmaxf(mag, fmaxf(fabsf(l), fabsf(h)));
    }
    const float rt = fast_sqrt(rt2) * 1.0001f + 1e-5f * mag;  // 1-ulp sqrt: the bounds carry 1e-4 of slack
    return leaf_mask_of_sphere(cull, S, lane, ct, rt, mag, lower);
}

// One wave = 256 consecutive points of one configuration per pass; all global traffic in contiguous 1 KB pieces
// through a wave-private LDS slice (same scheme as cached_query_wave, see cached.hip).
constexpr int kWavesPerBlock = 4;
constexpr int kTilePoints = 256;
// configurations per launch (a grid dimension carries them); a build knob so that a test can cross the slab border with
// a small batch
#ifndef PVAMD_COMPOSED_SLAB
#define PVAMD_COMPOSED_SLAB 65535
#endif
constexpr int kConfigSlab = PVAMD_COMPOSED_SLAB;
// fewer (tile, configuration) pairs than this: the one-point-per-lane kernel.  A tile is one wave's work and the chip
// holds 6,144-8,192 waves: below ~4 rounds of them the last, partly filled round costs more than the tile machinery saves
// (README case, 200 x 15,251 points = 12,000 tiles: per-lane 0.076 / 0.061 ms on 21 MB / 100 KB link grids against 0.104 /
// 0.088 ms; C4's 204,800 tiles: wave-tile 0.89 against 0.95 ms; tools/readme_case.py, profiles/r03_readme_case.txt)
#ifndef PVAMD_COMPOSED_WAVE_MIN_TILES
#define PVAMD_COMPOSED_WAVE_MIN_TILES 32768
#endif
constexpr int64_t kWaveTileMinTiles = PVAMD_COMPOSED_WAVE_MIN_TILES;
#ifndef PVAMD_COMPOSED_PPP
#define PVAMD_COMPOSED_PPP 2
#endif
// kEstimate (instruction-bound, L2-resident grids): 8 waves per SIMD (<= 64 VGPRs, a few spills) beat the 6 the allocator
// would pick on its own, C4 0.84 -> 0.80 ms.  kInlineExact (gather-bound, large grids): forcing 8 costs spills around the
// division sequence, 6.19 -> 6.57 ms on the README-size robot; the allocator's own choice (5-6) is left alone.
#ifndef PVAMD_COMPOSED_MINWAVES
#define PVAMD_COMPOSED_MINWAVES 8
#endif
// the inline-exact build: 6 = what the allocator chose on its own (78-80 VGPRs) until the switchable index rules added
// statements to the exact path (89); held there
#ifndef PVAMD_COMPOSED_MINWAVES_INLINE
#define PVAMD_COMPOSED_MINWAVES_INLINE 6
#endif

// The leaf loop of one tile: PPP points per lane at a time, results into the wave's LDS slice (or packed, to memory).
// MASKED = false: the tile was not worth a leaf mask (every leaf is visited; no bit tests, no refinement).
template <int PPP, int MODE, bool PACKED, bool MASKED>
PVAMD_DEV void tile_passes(const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A, int a,
                           int64_t first, int64_t P, float* __restrict__ val, int* __restrict__ leaf, float* spf, int lane,
                           uint64_t todo, float lower) {
    float* svf = spf + 768;
    const int first_leaf = todo ? __builtin_ctzll(todo) : 0;
    // A tile compact enough for the static test to drop a leaf is worth re-testing as the minimum tightens: after
    // every visited leaf the wave's largest running minimum is an upper bound of every point's final value, and the
    // leaves whose lower bound exceeds it are dropped (strictly greater, so ties cannot be affected).  Scattered
    // tiles (nothing dropped statically) skip the ~10 instructions per visited leaf.
    const bool refine = MASKED && S <= 64 && todo != (S >= 64 ? ~0ull : ((1ull << S) - 1ull));
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
            best[k] = best_init(first_leaf);
        }
        bool unsure[PPP];
#pragma unroll
        for (int k = 0; k < PPP; ++k) unsure[k] = false;
        uint64_t rem = todo;
        for (int s = 0; s < S; ++s) {
            if (MASKED && s < 64 && !((rem >> s) & 1ull)) continue;  // wave-uniform
            const float* M = tf + 16 * ((int64_t)s * A + a);  // wave-uniform: scalar loads
            const pvamd_grid_t& g = grids[s];
            // all PPP candidates first, their comparisons after: the gathers of the PPP points are in flight together
            float v[PPP], ga[PPP], gb[PPP], gc[PPP];
            uint64_t vm[PPP];
#pragma unroll
            for (int k = 0; k < PPP; ++k)
                leaf_candidate<MODE>(g, M, px[k], py[k], pz[k], v[k], ga[k], gb[k], gc[k], vm[k], unsure[k]);
#pragma unroll
            for (int k = 0; k < PPP; ++k)
                keep_first_minimum(best[k], s, v[k], ga[k], gb[k], gc[k], __builtin_amdgcn_inverse_ballot_w64(vm[k]));
            if (refine) {
                float m = best[0].v;
#pragma unroll
                for (int k = 1; k < PPP; ++k) m = __builtin_fmaxf(m, best[k].v);
                const float ub = wave_max(m);  // NaN minima are ignored: nothing replaces them anyway
                rem &= ~__builtin_amdgcn_ballot_w64(lower > ub + 1e-6f * fabsf(ub));
            }
        }
        if constexpr (MODE == kEstimate) {
            // the few points whose index estimate could not be trusted for some leaf: all over again, exactly
#pragma unroll
            for (int k = 0; k < PPP; ++k) {
                if (__builtin_expect(wave_any(unsure[k]), 0)) {
                    Best redo = best_init(first_leaf);
                    bool dummy = false;
                    walk_leaves<kExact>(grids, S, tf, A, a, todo, px[k], py[k], pz[k], redo, dummy);
                    if (unsure[k]) best[k] = redo;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PPP; ++k) {
            const int p = lane + 64 * (h + k);
            const int s_win = best[k].tag & (kUnnormalised - 1);
            // per-lane winner: these matrix reads are vector loads, but the S*A stack is tiny and cache-resident
            const float* M = tf + 16 * ((int64_t)s_win * A + a);
            float gx, gy, gz;
            rotate_back(M, best[k], gx, gy, gz);
            if constexpr (PACKED) {
                // one (val, gx, gy, gz) record per point, in processing order: lanes hold consecutive points, so the
                // wave's store is a contiguous 1 KB as it is; plain stores -- the un-permute pass reads them back
                // from L2 / Infinity Cache right away
                reinterpret_cast<f32x4*>(val)[(int64_t)a * P + first + p] = f32x4{best[k].v, gx, gy, gz};
            } else {
                svf[p] = best[k].v;  // a lane overwrites only the LDS slots of the points it owns
                spf[3 * p] = gx;
                spf[3 * p + 1] = gy;
                spf[3 * p + 2] = gz;
            }
            if (leaf) leaf[(int64_t)a * P + first + p] = s_win;
        }
    }
}


// ---- round 4: two running minima instead of one ----
// The round-3 loop keeps ONE first minimum over all candidates, so every visit pays an exact square root (11 vector
// instructions) to bring its bounding-box candidate into the domain of the cached values, and a five-register update.
// Here the OUT-OF-RANGE candidates have their own register minimum ordered by the SQUARED norm (sqrt is monotone: n2_b <
// n2_a decides, except where the two correctly rounded roots could coincide -- a band of 2^-21 relative, re-decided with
// both exact roots behind a wave-uniform branch; exact ties keep the incumbent as torch.argmin does) and the winner's root
// is taken once per point; the IN-RANGE candidates keep (value, leaf, flat index) -- three registers, the record is
// gathered once more for the winner -- and the two are compared by (value, leaf) at the end: the reference's first minimum
// over all leaves (sdf.py:421), bit for bit.
// Tried first and measured slower than the round-3 loop although they issue fewer vector instructions
// (profiles/r04_composed_variants.txt): the in-range (point, leaf) pairs compacted across lanes and visits in an LDS queue,
// drained 64 at a time with the leaf constants from per-lane global reads (v1) or an LDS table (v3), and a per-lane leaf
// bitmask walked after the loop (v2) -- the queue's drains and merges are chains of dependent LDS / memory round trips
// that the wave waits out (SQ_WAIT_ANY + 64 %), and the bitmask walk runs ~3 iterations at 2 live lanes.
#ifdef PVAMD_COMPOSED_STATS  // tools/band_rate.py: how often the exact-root band is entered (a variant build only)
__device__ unsigned long long g_band_stats[4];  // 64-point visits | visits that enter the band | lanes in the band | lanes the exact roots turn back
extern "C" int pvamd_debug_band_stats(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_band_stats), sizeof(g_band_stats));
    if (reset) { unsigned long long z[4] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_band_stats), z, sizeof(z)); }
    return 0;
}
#define BAND_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_band_stats[i], (unsigned long long)(v)); } while (0)
#else
#define BAND_STAT(i, v)
#endif
constexpr float kNearTie = 0.99999952316284179688f;  // 1 - 2^-21: sqrt_rn(n2_b) == sqrt_rn(n2_a) needs n2_b >= n2_a (1 - 2^-22)
constexpr int kNoLeaf = kUnnormalised - 1;           // "no candidate yet": loses every (value, leaf) tie

struct BestOut {
    float n2, tx, ty, tz;  // squared bounding-box distance and the (unnormalised) bounding-box vector, leaf frame
    int leaf;
};
struct BestIn {
    float v;
    int leaf, flat;
};

// GROUPED (round 6, composed_query_grouped below): the wave's 256 points are a spatially compact run of its workgroup's
// chunk, read from the chunk-sorted copy in memory (`pts`: lane stride 12 B, contiguous over the wave); a result goes to
// the LDS slot of the CALLER-order position `perm` names (tile = idx >> 8, point = idx & 255 of the block's `res` slices,
// the layout the coalesced stores read), so the un-permutation costs four LDS writes per point and no memory pass.
struct GroupIO {
    const float* __restrict__ pts;      // the wave's 256 sorted points
    const uint16_t* __restrict__ perm;  // their positions inside the chunk, caller order
    float* res;                         // the block's result slices, [tile][1024]
    int64_t chunk_first;                // caller index of the chunk's first point
    const uint16_t* lperm;              // GROUPED == 2 (composed_query_fused): the same positions, in LDS; the points are in `res`
};

// GROUPED: 0 = the wave's own tile; 1 = a sorted copy + positions in memory (group_points_kernel ran); 2 = the chunk was sorted by
// this very workgroup: positions in LDS, a point is read from the caller-order slot its result will overwrite
template <int PPP, bool PACKED, bool MASKED, int GROUPED = 0>
PVAMD_DEV void tile_passes_split(const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A, int a,
                                 int64_t first, int64_t P, float* __restrict__ val, int* __restrict__ leaf, float* spf,
                                 int lane, uint64_t todo, float lower, const GroupIO gio = GroupIO{}) {
    float* svf = spf + 768;
    const int first_leaf = todo ? __builtin_ctzll(todo) : 0;
    const bool refine = MASKED && S <= 64 && todo != ((S >= 64) ? ~0ull : ((1ull << S) - 1ull));
    const uint64_t everyone = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
    for (int h = 0; h < 4; h += PPP) {
        float px[PPP], py[PPP], pz[PPP];
        int slot_of[PPP];  // GROUPED == 2 only
        BestOut best[PPP];
        BestIn bin[PPP];
        uint64_t unsure[PPP];
#pragma unroll
        for (int k = 0; k < PPP; ++k) {
            const int p = lane + 64 * (h + k);
            if constexpr (GROUPED == 2) {
                slot_of[k] = gio.lperm[p];
                const float* q = gio.res + (slot_of[k] >> 8) * 1024 + 3 * (slot_of[k] & 255);
                px[k] = q[0];
                py[k] = q[1];
                pz[k] = q[2];
            } else if constexpr (GROUPED == 1) {
                px[k] = gio.pts[3 * p];
                py[k] = gio.pts[3 * p + 1];
                pz[k] = gio.pts[3 * p + 2];
            } else {
                px[k] = spf[3 * p];
                py[k] = spf[3 * p + 1];
                pz[k] = spf[3 * p + 2];
            }
            best[k] = BestOut{__builtin_inff(), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), kNoLeaf};
            bin[k] = BestIn{__builtin_inff(), kNoLeaf, 0};
            unsure[k] = 0;
        }
        uint64_t rem = todo;
        for (int s = 0; s < S; ++s) {
            if (MASKED && s < 64 && !((rem >> s) & 1ull)) continue;  // wave-uniform
            const float* M = tf + 16 * ((int64_t)s * A + a);  // wave-uniform: scalar loads
            const pvamd_grid_t& g = grids[s];
#pragma unroll
            for (int k = 0; k < PPP; ++k) {
                const float x = affine_row(M[0], M[1], M[2], M[3], px[k], py[k], pz[k]);
                const float y = affine_row(M[4], M[5], M[6], M[7], px[k], py[k], pz[k]);
                const float z = affine_row(M[8], M[9], M[10], M[11], px[k], py[k], pz[k]);
                // sdf.py:559-568 for every lane, squared (no exec masking; in-range lanes are masked out of the take)
                const float ta = __builtin_amdgcn_fmed3f(sub_rn(x, g.bb_min[0]), sub_rn(x, g.bb_max[0]), 0.f);
                const float tb = __builtin_amdgcn_fmed3f(sub_rn(y, g.bb_min[1]), sub_rn(y, g.bb_max[1]), 0.f);
                const float tc = __builtin_amdgcn_fmed3f(sub_rn(z, g.bb_min[2]), sub_rn(z, g.bb_max[2]), 0.f);
                const float n2 = fmaf(tc, tc, fmaf(tb, tb, mul_rn(ta, ta)));
                // Round 6: ONE compare stands in front of the range test.  Every statement from x to n2 is monotone in how far
                // x lies outside the box, so over the valid interval [vlo, vhi] n2 is largest at an end point:
                // pvamd_grid_finalize() evaluates these same statements there (range_n2), and n2 > range_n2 PROVES the
                // point out of range -- exactly, no tolerance.  With neighbouring points per wave (the grouped kernel) most
                // visits end here: 9 vector instructions and the vlo / vhi scalar loads become one compare.  A NaN n2 (NaN
                // point, or a descriptor without a box) proves nothing and takes the range test.
                uint64_t vm = 0;
                if (__builtin_amdgcn_ballot_w64(!(n2 > g.range_n2)) != 0) {
                    vm = in_range_mask(g, x, y, z);
                    if (vm != 0) {
                        // the lanes in range look their value up (index estimate; shaky ones are redone exactly below)
                        const bool valid = __builtin_amdgcn_inverse_ballot_w64(vm);
                        bool shaky = false;
                        const int flat = voxel_flat_estimate(g, x, y, z, shaky);
                        unsure[k] |= vm & __builtin_amdgcn_ballot_w64(shaky);
                        float v = __builtin_inff();
                        if (valid) v = reinterpret_cast<const float __attribute__((address_space(1)))*>((uintptr_t)g.vox)[4 * (int64_t)flat];
                        // ascending leaves: "strictly smaller, or the first" is the first minimum; NaN counts as the minimum
                        const uint64_t better = __builtin_amdgcn_ballot_w64(!(v >= bin[k].v)) & __builtin_amdgcn_ballot_w64(bin[k].v == bin[k].v);
                        const bool t = __builtin_amdgcn_inverse_ballot_w64(vm & (better | __builtin_amdgcn_ballot_w64(bin[k].leaf == kNoLeaf)));
                        bin[k].v = t ? v : bin[k].v;
                        bin[k].leaf = t ? s : bin[k].leaf;
                        bin[k].flat = t ? flat : bin[k].flat;
                        if (vm == everyone) continue;
                    }
                }
                // first minimum, NaN counts as minimum (keep_first_minimum), on the squared norms
                uint64_t take = __builtin_amdgcn_ballot_w64(!(n2 >= best[k].n2)) &
                                __builtin_amdgcn_ballot_w64(best[k].n2 == best[k].n2) & ~vm;
                BAND_STAT(0, 1);
                // nobody improves (neighbouring points agree on which leaves are far): the near-tie test and the five
                // selects are skipped for the wave
                if (take == 0) continue;
                const uint64_t near = take & __builtin_amdgcn_ballot_w64(n2 >= mul_rn(best[k].n2, kNearTie));
                if (__builtin_expect(near != 0, 0)) {
                    // the two roots may round to the same float32, in which case the incumbent stays: decide exactly
                    const float ra = sqrt_rn_sumsq(best[k].n2), rb = sqrt_rn_sumsq(n2);
                    BAND_STAT(1, 1); BAND_STAT(2, __popcll(near)); BAND_STAT(3, __popcll(near & __builtin_amdgcn_ballot_w64(!(rb < ra))));
                    take &= ~(near & __builtin_amdgcn_ballot_w64(!(rb < ra)));
                }
                const bool t = __builtin_amdgcn_inverse_ballot_w64(take);
                best[k].n2 = t ? n2 : best[k].n2;
                best[k].tx = t ? ta : best[k].tx;
                best[k].ty = t ? tb : best[k].ty;
                best[k].tz = t ? tc : best[k].tz;
                best[k].leaf = t ? s : best[k].leaf;
            }
            if (refine) {
                // every point's final value is <= its out-of-range minimum and <= its in-range minimum
                float m = -__builtin_inff();
#pragma unroll
                for (int k = 0; k < PPP; ++k)
                    m = __builtin_fmaxf(m, __builtin_fminf(fast_sqrt(best[k].n2) * 1.0001f + 1e-18f, bin[k].v));
                const float ub = wave_max(m);  // NaN minima are ignored: nothing replaces them anyway
                rem &= ~__builtin_amdgcn_ballot_w64(lower > ub + 1e-6f * fabsf(ub));
            }
        }
        Best fin[PPP];
#pragma unroll
        for (int k = 0; k < PPP; ++k) {
            // out-of-range minimum: the root once; no candidate at all (every leaf answered +inf) = the first visited leaf
            // with the NaN gradient, as the round-3 loop leaves it
            fin[k].v = sqrt_rn_sumsq(best[k].n2);
            fin[k].gx = best[k].tx;
            fin[k].gy = best[k].ty;
            fin[k].gz = best[k].tz;
            fin[k].tag = (best[k].leaf == kNoLeaf ? first_leaf : best[k].leaf) | kUnnormalised;
            const bool has = bin[k].leaf != kNoLeaf;
            if (wave_any(has)) {
                const int li = has ? bin[k].leaf : 0;
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has) r = load_record(grids[li].vox, bin[k].flat);  // per-lane descriptor read: S distinct addresses at most
                // (value, leaf) order with NaN as the smallest value: sdf.py:421 over all leaves
                const float vi = r.x, vo = fin[k].v;
                const bool vi_nan = vi != vi, vo_nan = vo != vo;
                const bool less = (vi_nan & !vo_nan) | (vi < vo);
                const bool same = (vi == vo) | (vi_nan & vo_nan);
                const bool wins = has & (less | (same & (li < best[k].leaf)));
                fin[k].v = wins ? vi : fin[k].v;
                fin[k].gx = wins ? r.y : fin[k].gx;
                fin[k].gy = wins ? r.z : fin[k].gy;
                fin[k].gz = wins ? r.w : fin[k].gz;
                fin[k].tag = wins ? li : fin[k].tag;
            }
        }
        // the few points whose index estimate could not be trusted for some leaf: all over again, exactly
#pragma unroll
        for (int k = 0; k < PPP; ++k) {
            if (__builtin_expect(unsure[k] != 0, 0)) {
                Best redo = best_init(first_leaf);
                bool dummy = false;
                walk_leaves<kExact>(grids, S, tf, A, a, todo, px[k], py[k], pz[k], redo, dummy);
                if (__builtin_amdgcn_inverse_ballot_w64(unsure[k])) fin[k] = redo;
            }
        }
#pragma unroll
        for (int k = 0; k < PPP; ++k) {
            const int p = lane + 64 * (h + k);
            const int s_win = fin[k].tag & (kUnnormalised - 1);
            const float* M = tf + 16 * ((int64_t)s_win * A + a);
            float gx, gy, gz;
            rotate_back(M, fin[k], gx, gy, gz);
            if constexpr (GROUPED != 0) {
                const int idx = GROUPED == 2 ? slot_of[k] : (int)gio.perm[p];
                float* slot = gio.res + (idx >> 8) * 1024;
                const int q = idx & 255;
                slot[768 + q] = fin[k].v;
                slot[3 * q] = gx;
                slot[3 * q + 1] = gy;
                slot[3 * q + 2] = gz;
                if (leaf) leaf[(int64_t)a * P + gio.chunk_first + idx] = s_win;
                continue;
            }
            if constexpr (PACKED) {
                reinterpret_cast<f32x4*>(val)[(int64_t)a * P + first + p] = f32x4{fin[k].v, gx, gy, gz};
            } else {
                svf[p] = fin[k].v;  // a lane overwrites only the LDS slots of the points it owns
                spf[3 * p] = gx;
                spf[3 * p + 1] = gy;
                spf[3 * p + 2] = gz;
            }
            if (leaf) leaf[(int64_t)a * P + first + p] = s_win;
        }
    }
}

// The same two minima for ONE point per lane (composed_query_scalar): all leaves, the exact index statements inline.
PVAMD_DEV Best walk_leaves_split(const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A, int a,
                                 float px, float py, float pz) {
    BestOut best{__builtin_inff(), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), kNoLeaf};
    BestIn bin{__builtin_inff(), kNoLeaf, 0};
    const uint64_t everyone = __builtin_amdgcn_ballot_w64(true);
    for (int s = 0; s < S; ++s) {
        const float* M = tf + 16 * ((int64_t)s * A + a);  // wave-uniform: scalar loads
        const pvamd_grid_t& g = grids[s];
        const float x = affine_row(M[0], M[1], M[2], M[3], px, py, pz);
        const float y = affine_row(M[4], M[5], M[6], M[7], px, py, pz);
        const float z = affine_row(M[8], M[9], M[10], M[11], px, py, pz);
        const uint64_t vm = in_range_mask(g, x, y, z);
        if (vm != 0) {
            const bool valid = __builtin_amdgcn_inverse_ballot_w64(vm);
            const int flat = voxel_flat_in_range_fused(g, x, y, z);
            float v = __builtin_inff();
            if (valid) v = reinterpret_cast<const float __attribute__((address_space(1)))*>((uintptr_t)g.vox)[4 * (int64_t)flat];
            const uint64_t better = __builtin_amdgcn_ballot_w64(!(v >= bin.v)) & __builtin_amdgcn_ballot_w64(bin.v == bin.v);
            const bool t = __builtin_amdgcn_inverse_ballot_w64(vm & (better | __builtin_amdgcn_ballot_w64(bin.leaf == kNoLeaf)));
            bin.v = t ? v : bin.v;
            bin.leaf = t ? s : bin.leaf;
            bin.flat = t ? flat : bin.flat;
            if (vm == everyone) continue;
        }
        const float ta = __builtin_amdgcn_fmed3f(sub_rn(x, g.bb_min[0]), sub_rn(x, g.bb_max[0]), 0.f);
        const float tb = __builtin_amdgcn_fmed3f(sub_rn(y, g.bb_min[1]), sub_rn(y, g.bb_max[1]), 0.f);
        const float tc = __builtin_amdgcn_fmed3f(sub_rn(z, g.bb_min[2]), sub_rn(z, g.bb_max[2]), 0.f);
        const float n2 = fmaf(tc, tc, fmaf(tb, tb, mul_rn(ta, ta)));
        uint64_t take = __builtin_amdgcn_ballot_w64(!(n2 >= best.n2)) & __builtin_amdgcn_ballot_w64(best.n2 == best.n2) & ~vm;
        const uint64_t near = take & __builtin_amdgcn_ballot_w64(n2 >= mul_rn(best.n2, kNearTie));
        if (__builtin_expect(near != 0, 0)) {
            const float ra = sqrt_rn_sumsq(best.n2), rb = sqrt_rn_sumsq(n2);
            take &= ~(near & __builtin_amdgcn_ballot_w64(!(rb < ra)));
        }
        const bool t = __builtin_amdgcn_inverse_ballot_w64(take);
        best.n2 = t ? n2 : best.n2;
        best.tx = t ? ta : best.tx;
        best.ty = t ? tb : best.ty;
        best.tz = t ? tc : best.tz;
        best.leaf = t ? s : best.leaf;
    }
    Best fin{sqrt_rn_sumsq(best.n2), best.tx, best.ty, best.tz, (best.leaf == kNoLeaf ? 0 : best.leaf) | kUnnormalised};
    const bool has = bin.leaf != kNoLeaf;
    if (wave_any(has)) {
        const int li = has ? bin.leaf : 0;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has) r = load_record(grids[li].vox, bin.flat);
        const float vi = r.x, vo = fin.v;
        const bool vi_nan = vi != vi, vo_nan = vo != vo;
        const bool less = (vi_nan & !vo_nan) | (vi < vo);
        const bool same = (vi == vo) | (vi_nan & vo_nan);
        const bool wins = has & (less | (same & (li < best.leaf)));
        fin.v = wins ? vi : fin.v;
        fin.gx = wins ? r.y : fin.gx;
        fin.gy = wins ? r.z : fin.gy;
        fin.gz = wins ? r.w : fin.gz;
        fin.tag = wins ? li : fin.tag;
    }
    return fin;
}

template <int PPP, int MODE, bool PACKED, int SPLIT>
__global__ __launch_bounds__(kWavesPerBlock * 64, MODE == kEstimate ? PVAMD_COMPOSED_MINWAVES : PVAMD_COMPOSED_MINWAVES_INLINE) void composed_query_wave(const pvamd_grid_t* __restrict__ grids, int S,
                                                                           const float* __restrict__ tf, int A,
                                                                           const float* __restrict__ pts,
                                                                           int64_t ntiles, int64_t P,
                                                                           float* __restrict__ val,
                                                                           float* __restrict__ grad,
                                                                           int* __restrict__ leaf, int a0) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][1024];
    __shared__ float cull[kMaxCullLeaves][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* spf = lds[wave];
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(spf);
    // blockIdx.x = configuration (fastest), blockIdx.y = tile block: the A configurations of one group of tiles run
    // back to back, so the leaf-grid region that tile touches (it moves little between configurations) and the tile's
    // points stay in L2 -- what matters once the grids are far larger than L2 (README-size link grids)
    const int a = a0 + blockIdx.x;
    build_cull_spheres(grids, S, tf, A, a, cull);
    __syncthreads();
    // The mask drops leaves only for tiles that are small against the scene, and costs ~150 instructions a tile (6 % of
    // C4 on random points, where it drops nothing: 0.834 -> 0.792 ms without it).  `scene` = radius about leaf 0's centre
    // that holds every leaf's range; a tile whose first and last point are further apart than kMaskSpan of it is not
    // worth a mask (always safe: no mask = every leaf).
    float scene = __builtin_inff();
    if (S <= kMaxCullLeaves) {
        float reach = -__builtin_inff();
        if (lane < S) {
            const float dx = cull[lane][0] - cull[0][0], dy = cull[lane][1] - cull[0][1], dz = cull[lane][2] - cull[0][2];
            reach = fast_sqrt(dx * dx + dy * dy + dz * dz) + cull[lane][3];
        }
        scene = wave_max(reach);
    }
    const float span2 = (kMaskSpan * scene) * (kMaskSpan * scene);
    const int64_t wstride = (int64_t)gridDim.y * kWavesPerBlock;
    for (int64_t tile = (int64_t)blockIdx.y * kWavesPerBlock + wave; tile < ntiles; tile += wstride) {
        // P is any count >= 256: the LAST tile is moved back to end at the last point, overlapping its neighbour, so that
        // every tile is whole (the overlap is computed twice and written twice with the same bits -- cheaper than a
        // partial-tile path, whose registers every tile would pay for).  The array is only known to be 4-byte aligned.
        const int64_t first = tile * kTilePoints <= P - kTilePoints ? tile * kTilePoints : P - kTilePoints;
        const f32x4_u* src = reinterpret_cast<const f32x4_u*>(pts + 3 * first);  // re-read for every configuration: L2-resident
        sp[lane] = src[lane];
        sp[lane + 64] = src[lane + 64];
        sp[lane + 128] = src[lane + 128];
        PVAMD_WAVE_SYNC();
        float lower;
#ifdef PVAMD_NO_TILE_MASK
        lower = -__builtin_inff();
        const uint64_t todo = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
        const bool masked = false;
#else
        uint64_t todo;
        bool masked;
        {
            const float ex = spf[765] - spf[0], ey = spf[766] - spf[1], ez = spf[767] - spf[2];  // last - first point: LDS broadcasts
            masked = ex * ex + ey * ey + ez * ez <= span2;  // false for a NaN, too
            if (masked) {
                todo = tile_leaf_mask(cull, S, lane, spf, lower);
            } else {
                lower = -__builtin_inff();
                todo = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
            }
        }
#endif
        // two copies of the leaf loop only where instructions are what binds (kEstimate: grids that live in L2); the
        // gather-bound kInlineExact build loses more to the larger body than the simpler loop gives (README-size robot,
        // sorted points: 1.00 -> 1.10 ms with both copies)
        if constexpr (SPLIT != 0) {
            if (masked) tile_passes_split<PPP, PACKED, true>(grids, S, tf, A, a, first, P, val, leaf, spf, lane, todo, lower);
            else tile_passes_split<PPP, PACKED, false>(grids, S, tf, A, a, first, P, val, leaf, spf, lane, todo, lower);
        } else {
            if (MODE != kEstimate || masked) tile_passes<PPP, MODE, PACKED, true>(grids, S, tf, A, a, first, P, val, leaf, spf, lane, todo, lower);
            else tile_passes<PPP, MODE, PACKED, false>(grids, S, tf, A, a, first, P, val, leaf, spf, lane, todo, lower);
        }
        PVAMD_WAVE_SYNC();
        if constexpr (PACKED) continue;
        // row a starts at a * P floats: 16-byte aligned only when P % 4 == 0 -- the stores take any dword address
        const int64_t o = (int64_t)a * P + first;
        __builtin_nontemporal_store(sp[192 + lane], reinterpret_cast<f32x4_u*>(val + o) + lane);
        f32x4_u* dst = reinterpret_cast<f32x4_u*>(grad + 3 * o);
        __builtin_nontemporal_store(sp[lane], dst + lane);
        __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
        __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128);
        PVAMD_WAVE_SYNC();
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void composed_query_scalar(const pvamd_grid_t* __restrict__ grids, int S,
                                                              const float* __restrict__ tf, int A,
                                                              const float* __restrict__ pts, int64_t first,
                                                              int64_t P, float* __restrict__ val,
                                                              float* __restrict__ grad, int* __restrict__ leaf, int a0,
                                                              int config_fastest) {
    // two block orders: points fastest (blockIdx.x = point block, blockIdx.y = configuration) or, like the wave-tile
    // kernel, configuration fastest (blockIdx.x = configuration): the configurations of one block of points run back to
    // back and share the grid lines those points touch
    const int a = a0 + (config_fastest ? blockIdx.x : blockIdx.y);
    const int64_t bx = config_fastest ? blockIdx.y : blockIdx.x, nbx = config_fastest ? gridDim.y : gridDim.x;
    const int64_t stride = nbx * blockDim.x;
    const int64_t n = P - first;
    // whole waves iterate together (leaf_candidate votes across the wave): lanes past the end carry a NaN point
    const int64_t rounds = (n + stride - 1) / stride;
    for (int64_t r = 0; r < rounds; ++r) {
        const int64_t i = first + r * stride + bx * blockDim.x + threadIdx.x;
        const bool live = i < P;
        const float nanv = __builtin_nanf("");
        const float px = live ? pts[3 * i] : nanv, py = live ? pts[3 * i + 1] : nanv, pz = live ? pts[3 * i + 2] : nanv;
        Best best = best_init(0);
        if constexpr (SPLIT) {
            best = walk_leaves_split(grids, S, tf, A, a, px, py, pz);
        } else {
            bool unsure = false;
            walk_leaves<kInlineExact>(grids, S, tf, A, a, ~0ull, px, py, pz, best, unsure);
        }
        if (live) {
            const int s_win = best.tag & (kUnnormalised - 1);
            float gx, gy, gz;
            rotate_back(tf + 16 * ((int64_t)s_win * A + a), best, gx, gy, gz);
            const int64_t o = (int64_t)a * P + i;
            // non-temporal: nothing on the device re-reads a result before the caller does (README A = 200 slice 0.0520 -> 0.0507 ms,
            // 1M-point single-configuration call 0.0257 -> 0.0250 ms; profiles/r06_cq_direct.txt section 5)
            __builtin_nontemporal_store(best.v, val + o);
            __builtin_nontemporal_store(gx, grad + 3 * o);
            __builtin_nontemporal_store(gy, grad + 3 * o + 1);
            __builtin_nontemporal_store(gz, grad + 3 * o + 2);
            if (leaf) leaf[o] = s_win;
        }
    }
}

// ---- float64 query points with a float64 transform stack (a RobotSDF over a float64 chain): sdf.py:399 transforms in
// float64, every leaf answers in the query dtype (sdf.py:545-547), sdf.py:409 rotates the float64 gradient back.  One
// (configuration, point) per lane, configuration fastest; 24 B read + 32 B written per pair, fp64 VALU: a correctness path
// for the dtype contract, not a tuned one. ----
__global__ __launch_bounds__(256) void composed_query_f64_kernel(const pvamd_grid_t* __restrict__ grids, int S,
                                                                  const double* __restrict__ tf, int A,
                                                                  const double* __restrict__ pts, int64_t P,
                                                                  double* __restrict__ val, double* __restrict__ grad,
                                                                  int* __restrict__ leaf) {
    const int a = blockIdx.x;
    const int64_t stride = (int64_t)gridDim.y * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < P; i += stride) {
        const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        double bv = 0.0, bg[3] = {0.0, 0.0, 0.0};
        int bs = -1;
        for (int s = 0; s < S; ++s) {
            const double* M = tf + 16 * ((int64_t)s * A + a);  // wave-uniform: scalar loads
            double x[3], v, g0, g1, g2;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                x[r] = __builtin_fma(M[4 * r + 2], p[2], __builtin_fma(M[4 * r + 1], p[1], M[4 * r] * p[0])) + M[4 * r + 3];
            cached_lookup_f64(grids[s], x, v, g0, g1, g2);
            const bool take = (bs < 0) || (v < bv) || (v != v && bv == bv);  // sdf.py:421: first minimum, NaN counts as minimum
            if (take) {
                bv = v;
                bs = s;
#pragma unroll
                for (int j = 0; j < 3; ++j) bg[j] = __builtin_fma(M[8 + j], g2, __builtin_fma(M[4 + j], g1, M[j] * g0));
            }
        }
        const int64_t o = (int64_t)a * P + i;
        val[o] = bv;
        grad[3 * o] = bg[0];
        grad[3 * o + 1] = bg[1];
        grad[3 * o + 2] = bg[2];
        if (leaf) leaf[o] = bs;
    }
}

// ---- generic leaves (MeshSDF, SphereSDF, nested compositions): the glue of sdf.py:392-433 around per-leaf queries ----
// x[a][p] = T[a] p with the fused kernel's rounding (k-ordered fma chain, sdf.py:399)
__global__ __launch_bounds__(256) void transform_points_kernel(const float* __restrict__ tf, int A,
                                                                const float* __restrict__ pts, int64_t P,
                                                                float* __restrict__ out) {
    const int a = blockIdx.y;
    const float* M = tf + 16 * (int64_t)a;  // wave-uniform: scalar loads
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        float* o = out + 3 * ((int64_t)a * P + i);
        o[0] = affine_row(M[0], M[1], M[2], M[3], px, py, pz);
        o[1] = affine_row(M[4], M[5], M[6], M[7], px, py, pz);
        o[2] = affine_row(M[8], M[9], M[10], M[11], px, py, pz);
    }
}

// fold leaf s into the running first-minimum: g_obj = L^T g_leaf (sdf.py:409; L = linear part of obj->leaf, any affine
// transform), take where the leaf's value is smaller or is NaN against a number (sdf.py:421 argmin); first = 1 for s = 0
__global__ __launch_bounds__(256) void compose_merge_kernel(const float* __restrict__ tf, int A, int64_t P,
                                                             const float* __restrict__ leaf_val,
                                                             const float* __restrict__ leaf_grad, int s, int first,
                                                             float* __restrict__ best_val, float* __restrict__ best_grad,
                                                             int* __restrict__ best_leaf) {
    const int a = blockIdx.y;
    const float* M = tf + 16 * (int64_t)a;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const int64_t o = (int64_t)a * P + i;
        const float v = leaf_val[o], bv = first ? 0.f : best_val[o];
        const bool take = first || (v < bv) || (v != v && bv == bv);
        if (take) {
            const float gx = leaf_grad[3 * o], gy = leaf_grad[3 * o + 1], gz = leaf_grad[3 * o + 2];
            best_val[o] = v;
            best_grad[3 * o] = fmaf(M[8], gz, fmaf(M[4], gy, mul_rn(M[0], gx)));
            best_grad[3 * o + 1] = fmaf(M[9], gz, fmaf(M[5], gy, mul_rn(M[1], gx)));
            best_grad[3 * o + 2] = fmaf(M[10], gz, fmaf(M[6], gy, mul_rn(M[2], gx)));
            if (best_leaf) best_leaf[o] = s;
        }
    }
}

// tiles per wave in the un-permute pass (= gathers in flight per lane / 4): 1 is best (whole README-size call 1.65 ms;
// 1.70 with 2, 1.83 with 4 -- the pass is not latency-bound)
#ifndef PVAMD_UNPERMUTE_TILES
#define PVAMD_UNPERMUTE_TILES 1
#endif
constexpr int kUnpermuteTiles = PVAMD_UNPERMUTE_TILES;
// waves per workgroup of the un-permute pass: its own knob (the query kernel's 4 are tuned for its register budget)
#ifndef PVAMD_UNPERMUTE_WAVES
#define PVAMD_UNPERMUTE_WAVES 4
#endif
constexpr int kUnpermuteWaves = PVAMD_UNPERMUTE_WAVES;
// ---- bucketed path: un-permute ----
// The kernel above ran on spatially sorted points and left one packed record per (configuration, sorted position);
// this pass brings them back to the caller's point order: out[a][j] = packed[a][inv[j]].  One wave = 256 consecutive
// caller points of one configuration: 4 gathers of a 16-byte record per lane (a configuration's records are a 4 MB
// window that was written moments ago), results staged through the wave's LDS slice and written as 1 + 3 contiguous
// 1 KB stores like everywhere else.  Blocks of one configuration are kept on one XCD (block b runs on XCD b % 8) so that
// the window is pulled into ONE L2 instead of eight (README-size C4, whole bucketed call: 1.67 ms pinned, 2.06 ms with
// configuration-major blocks, 2.30 ms with configuration-fastest blocks).
__global__ __launch_bounds__(kUnpermuteWaves * 64) void composed_unpermute_kernel(const f32x4* __restrict__ packed,
                                                                                 const int* __restrict__ inv, int64_t P,
                                                                                 int64_t Pp, int A, int64_t tile_blocks,
                                                                                 float* __restrict__ val,
                                                                                 float* __restrict__ grad) {
    __shared__ __attribute__((aligned(16))) float lds[kUnpermuteWaves][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* spf = lds[wave];
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(spf);
    float* svf = spf + 768;
    // b -> (configuration, tile block): eight consecutive blocks go to the eight XCDs and carry eight different
    // configurations; the next eight continue the same configurations with the next tile block
    const int64_t b = blockIdx.x;
    const int64_t group = b / (8 * tile_blocks), within = b % (8 * tile_blocks);
    const int a = (int)(group * 8 + within % 8);
    if (a >= A) return;
    const f32x4* src = packed + (int64_t)a * Pp;
    // kUnpermuteTiles tiles per wave: all their gathers are issued before the first result is used
    f32x4 r[kUnpermuteTiles][4];
    const int64_t tile0 = ((within / 8) * kUnpermuteWaves + wave) * kUnpermuteTiles;
#pragma unroll
    for (int t = 0; t < kUnpermuteTiles; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = (tile0 + t) * kTilePoints + lane + 64 * k;
            r[t][k] = j < P ? src[inv[j]] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int t = 0; t < kUnpermuteTiles; ++t) {
        const int64_t j0 = (tile0 + t) * kTilePoints;
        if (j0 >= P) break;
        const int64_t o = (int64_t)a * P + j0;
        if (j0 + kTilePoints <= P) {  // rows of an odd P start at any dword: 16-byte stores take that (common.h)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = lane + 64 * k;
                svf[p] = r[t][k].x;
                spf[3 * p] = r[t][k].y;
                spf[3 * p + 1] = r[t][k].z;
                spf[3 * p + 2] = r[t][k].w;
            }
            PVAMD_WAVE_SYNC();
            __builtin_nontemporal_store(sp[192 + lane], reinterpret_cast<f32x4_u*>(val + o) + lane);
            f32x4_u* dst = reinterpret_cast<f32x4_u*>(grad + 3 * o);
            __builtin_nontemporal_store(sp[lane], dst + lane);
            __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
            __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128);
            PVAMD_WAVE_SYNC();
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t j = j0 + lane + 64 * k;
                if (j < P) {
                    val[(int64_t)a * P + j] = r[t][k].x;
                    float* g = grad + 3 * ((int64_t)a * P + j);
                    g[0] = r[t][k].y;
                    g[1] = r[t][k].z;
                    g[2] = r[t][k].w;
                }
            }
        }
    }
}


// ---- round 6: chunk-grouped points ----
// On random query points nearly every wave pays BOTH halves of nearly every leaf visit: a leaf's range covers 1.5 % (C4) / 5 %
// (C3) of the query box, so some of a wave's 64 scattered lanes are inside it 62-96 % of the time, and the look-up half then
// runs for a handful of lanes.  A global spatial sort fixes that (0.53 ms against 0.68 on C4) but its un-permute pass over the
// 839 MB of outputs costs more than it saves.  Regrouping the points INSIDE chunks of consecutive caller points keeps nearly
// all of the gain (tools/chunk_sort_probe.py: chunks of 4096 run the kernel 30 % faster, as fast as the global sort) and a
// workgroup can undo it through LDS: a chunk's outputs are one contiguous piece of every output row.
//
// group_points_kernel (once per call, shared by the A configurations): one workgroup per chunk of kGroupWaves x 256
// consecutive points sorts them along a Hilbert curve over the chunk's own box (16^3 cells, counting sort in LDS) and writes
// the sorted copy, `perm[j]` = position inside the chunk of sorted point j, and per run of 256 sorted points (one wave's
// tile) the bounding sphere the leaf mask wants.  composed_query_grouped: a wave takes one such run, every result lands in
// the LDS slot of its caller-order position, and after a barrier each wave stores one caller-order tile as 1 + 3 contiguous KB.
// The order inside a Hilbert cell is whatever the LDS atomics gave: no result depends on it (a point's statements do not
// know its lane).
#ifndef PVAMD_GROUP_WAVES
#define PVAMD_GROUP_WAVES 16
#endif
constexpr int kGroupWaves = PVAMD_GROUP_WAVES;
#ifndef PVAMD_FUSED_COHERENT_SPAN
#define PVAMD_FUSED_COHERENT_SPAN 0.45f
#endif
constexpr float kCoherentSpan = PVAMD_FUSED_COHERENT_SPAN;  // a chunk whose tiles each span at most this much of it is not sorted
constexpr int kGroupChunk = kGroupWaves * kTilePoints;
#ifndef PVAMD_FUSED_WAVES
#define PVAMD_FUSED_WAVES 16
#endif
constexpr int kFusedWaves = PVAMD_FUSED_WAVES;  // waves (x 256 points) per chunk of the in-workgroup sort
constexpr int kFusedChunk = kFusedWaves * kTilePoints;
#ifndef PVAMD_FUSED_MIN_BLOCKS
#define PVAMD_FUSED_MIN_BLOCKS 1024
#endif
constexpr int64_t kFusedMinBlocks = PVAMD_FUSED_MIN_BLOCKS;  // (chunk, configuration) workgroups below which the per-lane / wave-tile kernels keep the call
static_assert(kGroupChunk <= 65536 && 4096 % (kGroupWaves * 64) == 0, "perm is uint16; the scan splits 4096 bins evenly");

template <int NW>
__global__ __launch_bounds__(NW * 64) void group_points_kernel(const float* __restrict__ pts, int64_t P,
                                                               float* __restrict__ sorted, float* __restrict__ bounds,
                                                               uint16_t* __restrict__ perm) {
    constexpr int N = NW * kTilePoints, kBins = 4096, kPerThread = kBins / (NW * 64);
    __shared__ __attribute__((aligned(16))) float lds[NW][768];
    __shared__ unsigned hist[kBins];
    __shared__ uint16_t sperm[N];
    __shared__ unsigned box[6];
    __shared__ unsigned wsum[NW];
    __shared__ float wspan[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = blockIdx.x;
    const int64_t cfirst = chunk * N <= P - N ? chunk * N : P - N;  // the last chunk is moved back to end at the last point
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(lds[wave]);
    {
        const f32x4_u* src = reinterpret_cast<const f32x4_u*>(pts + 3 * (cfirst + wave * kTilePoints));
        sp[lane] = src[lane];
        sp[lane + 64] = src[lane + 64];
        sp[lane + 128] = src[lane + 128];
    }
    for (int i = threadIdx.x; i < kBins; i += NW * 64) hist[i] = 0u;
    if (threadIdx.x < 3) {
        box[threadIdx.x] = 0xffffffffu;
        box[3 + threadIdx.x] = 0u;
    }
    PVAMD_WAVE_SYNC();
    const f32x4 q0 = sp[3 * lane], q1 = sp[3 * lane + 1], q2 = sp[3 * lane + 2];
    const float px[4] = {q0.x, q0.w, q1.z, q2.y}, py[4] = {q0.y, q1.x, q1.w, q2.z}, pz[4] = {q0.z, q1.y, q2.x, q2.w};
    // the chunk's box over its finite coordinates
    float lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = __builtin_inff();
        hi[d] = -__builtin_inff();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float c[3] = {px[k], py[k], pz[k]};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const bool fin = fabsf(c[d]) < __builtin_inff();
            lo[d] = fin ? fminf(lo[d], c[d]) : lo[d];
            hi[d] = fin ? fmaxf(hi[d], c[d]) : hi[d];
        }
    }
    __syncthreads();  // box[] initialised, every tile in LDS
    // L1 diameter of the widest run of 64 consecutive caller-order points of this wave (a lane holds four consecutive points, so a
    // run is a row of 16 lanes): what the leaf loop's 64 lanes would be handed without regrouping
    float run_span = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) run_span += fmaxf(group16_max(hi[d]) - group16_min(lo[d]), 0.f);
    const float tile_span = wave_max(run_span);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float l = wave_min(lo[d]), h = wave_max(hi[d]);
        if (lane == 0) {
            atomicMin(&box[d], order_code(l));
            atomicMax(&box[3 + d], order_code(h));
        }
    }
    if (lane == 0) wspan[wave] = tile_span;
    __syncthreads();
    float blo[3], scale[3], chunk_span = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        blo[d] = order_decode(box[d]);
        const float ext = order_decode(box[3 + d]) - blo[d];
        chunk_span += fmaxf(ext, 0.f);
        scale[d] = 15.999f / fmaxf(ext, 1e-30f);
    }
    // a chunk whose runs of 64 consecutive points each span at most kCoherentSpan of it is already coherent where it matters -- the 64
    // lanes of a leaf-loop pass -- (an ordered slice, a pre-sorted set): it keeps the caller's order; regrouping a 512 x 512 slice
    // made C4 8 % SLOWER (the LDS scatter of the results costs, and there was nothing left to gain)
    float worst_tile = 0.f;
    for (int w = 0; w < NW; ++w) worst_tile = fmaxf(worst_tile, wspan[w]);
    const bool coherent = worst_tile <= kCoherentSpan * chunk_span;  // block-uniform
    unsigned key[4], rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        key[k] = hilbert_cell16(px[k], py[k], pz[k], blo, scale);  // NaN / infinite coordinates land in some cell: harmless
        rank[k] = atomicAdd(&hist[key[k]], 1u);
    }
    __syncthreads();
    // exclusive scan of the 4096 counts: kPerThread consecutive bins per thread, a wave scan, the wave totals
    unsigned local[kPerThread], sum = 0u;
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
        local[i] = hist[threadIdx.x * kPerThread + i];
        sum += local[i];
    }
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc, off);
        inc += lane >= off ? t : 0u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned base = inc - sum;
    for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
        hist[threadIdx.x * kPerThread + i] = base;
        base += local[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) sperm[hist[key[k]] + rank[k]] = (uint16_t)(wave * kTilePoints + 4 * lane + k);
    __syncthreads();
    if (coherent) {
        for (int k = 0; k < 4; ++k) sperm[wave * kTilePoints + 4 * lane + k] = (uint16_t)(wave * kTilePoints + 4 * lane + k);
        __syncthreads();
    }
    // wave w writes sorted run w and its bounding sphere (the statements of tile_leaf_mask)
    float tlo[3], thi[3];
    bool odd = false;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        tlo[d] = __builtin_inff();
        thi[d] = -__builtin_inff();
    }
    const int64_t out0 = chunk * N + wave * kTilePoints;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        const int idx = sperm[wave * kTilePoints + j];
        const float* q = &lds[idx >> 8][3 * (idx & 255)];
        perm[out0 + j] = (uint16_t)idx;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float x = q[d];
            sorted[3 * (out0 + j) + d] = x;
            tlo[d] = fminf(tlo[d], x);
            thi[d] = fmaxf(thi[d], x);
            odd |= !(fabsf(x) < __builtin_inff());
        }
    }
    float ct[3], rt2 = 0.f, mag = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float l = wave_min(tlo[d]), h = wave_max(thi[d]);
        ct[d] = 0.5f * (l + h);
        const float half = 0.5f * (h - l);
        rt2 += half * half;
        mag = fmaxf(mag, fmaxf(fabsf(l), fabsf(h)));
    }
    float rt = fast_sqrt(rt2) * 1.0001f + 1e-5f * mag;
    if (wave_any(odd)) rt = __builtin_inff();  // a NaN / infinite coordinate: no bounds, every leaf is visited
    if (lane < 8) {
        const float rec[8] = {ct[0], ct[1], ct[2], rt, mag, 0.f, 0.f, 0.f};
        float v = rec[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) v = lane == i ? rec[i] : v;
        bounds[8 * (chunk * NW + wave) + lane] = v;
    }
}

// how large a run's sphere may be against the scene (the radius about leaf 0's centre that holds every leaf's range) for a leaf
// mask to be worth its ~60 instructions
#ifndef PVAMD_GROUP_MASK_SPAN
#define PVAMD_GROUP_MASK_SPAN 0.5f
#endif
#ifndef PVAMD_GROUP_MINWAVES
#define PVAMD_GROUP_MINWAVES 8
#endif
template <int NW, int PPP, bool PACKED>
__global__ __launch_bounds__(NW * 64, PVAMD_GROUP_MINWAVES) void composed_query_grouped(
    const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A, const float* __restrict__ sorted,
    const float* __restrict__ bounds, const uint16_t* __restrict__ perm, int64_t nchunks, int64_t P, float* __restrict__ val,
    float* __restrict__ grad, int* __restrict__ leaf, int a0) {
    constexpr int N = NW * kTilePoints;
    __shared__ __attribute__((aligned(16))) float res[NW][1024];
    __shared__ float cull[kMaxCullLeaves][8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a = a0 + blockIdx.x;  // configuration fastest, as composed_query_wave
    build_cull_spheres(grids, S, tf, A, a, cull);
    __syncthreads();
    float scene = __builtin_inff();
    if (S <= kMaxCullLeaves) {
        float reach = -__builtin_inff();
        if (lane < S) {
            const float dx = cull[lane][0] - cull[0][0], dy = cull[lane][1] - cull[0][1], dz = cull[lane][2] - cull[0][2];
            reach = fast_sqrt(dx * dx + dy * dy + dz * dz) + cull[lane][3];
        }
        scene = wave_max(reach);
    }
    const float span = PVAMD_GROUP_MASK_SPAN * scene;
    for (int64_t chunk = blockIdx.y; chunk < nchunks; chunk += gridDim.y) {
        const int64_t cfirst = chunk * N <= P - N ? chunk * N : P - N;
        const int64_t run = chunk * N + wave * kTilePoints;
        const float* tb = bounds + 8 * (chunk * NW + wave);  // wave-uniform: scalar loads
        const float ct[3] = {tb[0], tb[1], tb[2]};
        const float rt = tb[3], mag = tb[4];
        float lower = -__builtin_inff();
        uint64_t todo = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
        const bool masked = 2.f * rt <= span;  // false for the "no bounds" marker (+inf)
        if (masked) todo = leaf_mask_of_sphere(cull, S, lane, ct, rt, mag, lower);
        const GroupIO gio{sorted + 3 * run, perm + run, &res[0][0], cfirst, nullptr};
        if (masked) tile_passes_split<PPP, false, true, 1>(grids, S, tf, A, a, 0, P, val, leaf, nullptr, lane, todo, lower, gio);
        else tile_passes_split<PPP, false, false, 1>(grids, S, tf, A, a, 0, P, val, leaf, nullptr, lane, todo, lower, gio);
        __syncthreads();  // every result of the chunk is in its caller-order slot
        const f32x4_alias* sp = reinterpret_cast<const f32x4_alias*>(res[wave]);
        const int64_t o = (int64_t)a * P + cfirst + wave * kTilePoints;
        if constexpr (PACKED) {
            // one (val, gx, gy, gz) record per point (what a query sharded over GPUs gathers): the wave's caller-order tile is
            // 4 KB of consecutive records; plain stores -- the all-gather / unpack reads them next
            const float* rf = res[wave];
            f32x4* rec = reinterpret_cast<f32x4*>(val) + o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = lane + 64 * k;
                rec[p] = f32x4{rf[768 + p], rf[3 * p], rf[3 * p + 1], rf[3 * p + 2]};
            }
        } else {
            __builtin_nontemporal_store(sp[192 + lane], reinterpret_cast<f32x4_u*>(val + o) + lane);
            f32x4_u* dst = reinterpret_cast<f32x4_u*>(grad + 3 * o);
            __builtin_nontemporal_store(sp[lane], dst + lane);
            __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
            __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128);
        }
        __syncthreads();  // the slices are free for the next chunk
    }
}


// ---- the same regrouping with NO pre-pass and no scratch: the workgroup sorts its chunk itself ----
// A single configuration (C3: 8 drills, 4M points) cannot amortise group_points_kernel's extra pass over the points (48 MB read,
// 56 MB written, 56 MB read again against 112 MB of compulsory traffic), and a caller of pvamd_composed_query brings no scratch.
// Here the chunk's points are loaded once into the LDS slices their results will leave from, sorted in place (Hilbert cell keys
// over the chunk's box, counting sort: the count table borrows the value quarter of the slices, which nothing touches until
// results arrive), and a wave then reads the points of its run through the position table.  ~250 instructions and six barriers
// per thread against ~1500 of leaf loop (S = 8): worth it wherever the wave-tile kernel ran on scattered points.
template <int NW, int PPP>
__global__ __launch_bounds__(NW * 64, PVAMD_GROUP_MINWAVES) void composed_query_fused(
    const pvamd_grid_t* __restrict__ grids, int S, const float* __restrict__ tf, int A, const float* __restrict__ pts, int64_t nchunks,
    int64_t P, float* __restrict__ val, float* __restrict__ grad, int* __restrict__ leaf, int a0) {
    // one count per Hilbert cell of the 16^3 grid, or per run of 2 / 4 consecutive cells of the curve when the chunk has fewer value
    // slots than cells (the table lives in the value quarters of the slices)
    constexpr int N = NW * kTilePoints, kBins = N < 4096 ? N : 4096, kPerThread = kBins / (NW * 64), kKeyShift = kBins == 4096 ? 0 : (kBins == 2048 ? 1 : 2);
    static_assert(kBins >= 1024 && kBins % (NW * 64) == 0, "4, 8 or 16 waves");
    __shared__ __attribute__((aligned(16))) float res[NW][1024];
    __shared__ uint16_t sperm[N];
    __shared__ unsigned box[6];
    __shared__ unsigned wsum[NW];
    __shared__ __attribute__((aligned(16))) float wspan[NW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a = a0 + blockIdx.x;
    auto hist = [&](int bin) -> unsigned& { return reinterpret_cast<unsigned*>(res[bin >> 8])[768 + (bin & 255)]; };
    // (The cells span the chunk's own box.  Cells over the SCENE instead -- the cube that holds every leaf's range, known from the
    // culling spheres without a reduction or a barrier -- were tried: C4 with the sort per configuration 0.651 -> 0.615 ms, but C3
    // 0.076 -> 0.088 ms, no better than unsorted: 16 cells per axis spread over empty space leave runs of 64 points long and thin.
    // The Z curve in place of the Hilbert table: slower on both.  profiles/r06_composed_variants.txt.)
    for (int64_t chunk = blockIdx.y; chunk < nchunks; chunk += gridDim.y) {
        const int64_t cfirst = chunk * N <= P - N ? chunk * N : P - N;
        f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(res[wave]);
        {
            const f32x4_u* src = reinterpret_cast<const f32x4_u*>(pts + 3 * (cfirst + wave * kTilePoints));
            sp[lane] = src[lane];
            sp[lane + 64] = src[lane + 64];
            sp[lane + 128] = src[lane + 128];
        }
        for (int i = threadIdx.x; i < kBins; i += NW * 64) hist(i) = 0u;
        if (threadIdx.x < 3) {
            box[threadIdx.x] = 0xffffffffu;
            box[3 + threadIdx.x] = 0u;
        }
        PVAMD_WAVE_SYNC();
        const f32x4 q0 = sp[3 * lane], q1 = sp[3 * lane + 1], q2 = sp[3 * lane + 2];
        const float px[4] = {q0.x, q0.w, q1.z, q2.y}, py[4] = {q0.y, q1.x, q1.w, q2.z}, pz[4] = {q0.z, q1.y, q2.x, q2.w};
        float lo[3], hi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = __builtin_inff();
            hi[d] = -__builtin_inff();
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c[3] = {px[k], py[k], pz[k]};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const bool fin = fabsf(c[d]) < __builtin_inff();
                lo[d] = fin ? fminf(lo[d], c[d]) : lo[d];
                hi[d] = fin ? fmaxf(hi[d], c[d]) : hi[d];
            }
        }
        __syncthreads();  // box[] and the count table initialised, every tile in LDS
        // L1 diameter of this wave's 256 points.  (The pre-pass kernel measures runs of 64 instead -- the finer test; here its extra
        // reductions cost the scattered case 2 % and the tile test already catches ordered slices.)
        float tile_span = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float l = wave_min(lo[d]), h = wave_max(hi[d]);
            tile_span += fmaxf(h - l, 0.f);
            if (lane == 0) {
                atomicMin(&box[d], order_code(l));
                atomicMax(&box[3 + d], order_code(h));
            }
        }
        if (lane == 0) wspan[wave] = tile_span;
        __syncthreads();
        float blo[3], scale[3], chunk_span = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            blo[d] = order_decode(box[d]);
            const float ext = order_decode(box[3 + d]) - blo[d];
            chunk_span += fmaxf(ext, 0.f);
            scale[d] = 15.999f / fmaxf(ext, 1e-30f);
        }
        // The caller's order may already be coherent (an ordered slice, a pre-sorted set): where every wave's tile spans at most
        // kCoherentSpan of the chunk, every wave keeps its own tile and the sort -- a quarter of this kernel -- is skipped for the
        // chunk.  Block-uniform: every thread reads the same 16 spans.
        float worst_tile = 0.f;
        if constexpr (NW % 4 == 0) {  // four spans per LDS read
            const f32x4_alias* w4 = reinterpret_cast<const f32x4_alias*>(wspan);
#pragma unroll
            for (int w = 0; w < NW / 4; ++w) {
                const f32x4 v = w4[w];
                worst_tile = fmaxf(worst_tile, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
        } else {
            for (int w = 0; w < NW; ++w) worst_tile = fmaxf(worst_tile, wspan[w]);
        }
        const bool coherent = worst_tile <= kCoherentSpan * chunk_span;
        if (coherent) {
            // identity positions (each wave fills and later reads only its own 256 entries): the same leaf loop, no second copy of it
#pragma unroll
            for (int k = 0; k < 4; ++k) sperm[wave * kTilePoints + 4 * lane + k] = (uint16_t)(wave * kTilePoints + 4 * lane + k);
            PVAMD_WAVE_SYNC();
        } else {
        unsigned key[4], rank[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            key[k] = hilbert_cell16(px[k], py[k], pz[k], blo, scale) >> kKeyShift;
            rank[k] = atomicAdd(&hist((int)key[k]), 1u);
        }
        __syncthreads();
        unsigned local[kPerThread], sum = 0u;
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            local[i] = hist((int)threadIdx.x * kPerThread + i);
            sum += local[i];
        }
        unsigned inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(inc, off);
            inc += lane >= off ? t : 0u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned base = inc - sum;
        for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            hist((int)threadIdx.x * kPerThread + i) = base;
            base += local[i];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) sperm[hist((int)key[k]) + rank[k]] = (uint16_t)(wave * kTilePoints + 4 * lane + k);
        __syncthreads();  // the position table is complete; the count table is dead: the value quarters are free for results
        }
        const GroupIO gio{nullptr, nullptr, &res[0][0], cfirst, sperm + wave * kTilePoints};
        const uint64_t todo = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
        tile_passes_split<PPP, false, false, 2>(grids, S, tf, A, a, 0, P, val, leaf, nullptr, lane, todo, -__builtin_inff(), gio);
        __syncthreads();  // every result of the chunk is in its caller-order slot
        const int64_t o = (int64_t)a * P + cfirst + wave * kTilePoints;
        __builtin_nontemporal_store(sp[192 + lane], reinterpret_cast<f32x4_u*>(val + o) + lane);
        f32x4_u* dst = reinterpret_cast<f32x4_u*>(grad + 3 * o);
        __builtin_nontemporal_store(sp[lane], dst + lane);
        __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
        __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128);
        __syncthreads();  // the slices are free for the next chunk
    }
}

}  // namespace pvamd

using namespace pvamd;

// the round-3 leaf loop (one register minimum over all candidates, exact roots inside the loop): on request
static inline bool legacy_leaf_loop(int32_t flags, int32_t S) { return (flags & PVAMD_COMPOSED_LEGACY_LEAF_LOOP) || S >= kNoLeaf; }

extern "C" int pvamd_composed_query_packed(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A,
                                           const float* points, int64_t Pp, float* out_rec, int32_t flags, void* stream) {
    if (S < 1 || A < 1 || Pp < 1 || Pp % kTilePoints != 0 || S >= kUnnormalised) return PVAMD_E_SHAPE;
    if (!grids || !tf || !points || !out_rec) return PVAMD_E_NULL;
    if (!aligned_to(grids, 8) || !aligned_to(tf, 4) || !aligned_to(points, 16) || !aligned_to(out_rec, 16)) return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int64_t ntiles = Pp / kTilePoints;
    const int64_t tile_blocks = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    // gridDim.y of the query kernel = tile blocks (the configuration is blockIdx.x: any A)
    if (tile_blocks > 65535) return PVAMD_E_SHAPE;  // > 67 M points per call: use the direct entry point
    if (flags & PVAMD_COMPOSED_INLINE_EXACT)
        hipLaunchKernelGGL((composed_query_wave<PVAMD_COMPOSED_PPP, kInlineExact, true, 0>), dim3(A, (unsigned)tile_blocks),
                           dim3(kWavesPerBlock * 64), 0, s, grids, S, tf, A, points, ntiles, Pp, out_rec, nullptr, nullptr, 0);
    else if (legacy_leaf_loop(flags, S))
        hipLaunchKernelGGL((composed_query_wave<PVAMD_COMPOSED_PPP, kEstimate, true, 0>), dim3(A, (unsigned)tile_blocks),
                           dim3(kWavesPerBlock * 64), 0, s, grids, S, tf, A, points, ntiles, Pp, out_rec, nullptr, nullptr, 0);
    else
        hipLaunchKernelGGL((composed_query_wave<PVAMD_COMPOSED_PPP, kEstimate, true, 1>), dim3(A, (unsigned)tile_blocks),
                           dim3(kWavesPerBlock * 64), 0, s, grids, S, tf, A, points, ntiles, Pp, out_rec, nullptr, nullptr, 0);
    return (int)hipGetLastError();
}

extern "C" int pvamd_unpack_records(const float* rec, const int32_t* index, int64_t P, int64_t stride, int32_t A,
                                    float* out_val, float* out_grad, void* stream) {
    if (A < 1 || P < 1 || stride < 1) return PVAMD_E_SHAPE;
    if (!rec || !index || !out_val || !out_grad) return PVAMD_E_NULL;
    if (!aligned_to(rec, 16) || !aligned_to(index, 4) || !aligned_to(out_val, 4) || !aligned_to(out_grad, 4)) return PVAMD_E_ALIGN;
    const int64_t ntiles = (P + kTilePoints - 1) / kTilePoints;
    const int64_t tile_blocks = (ntiles + kUnpermuteWaves - 1) / kUnpermuteWaves;
    const int64_t groups = ((int64_t)A + 7) / 8;
    const int64_t ublocks = (tile_blocks + kUnpermuteTiles - 1) / kUnpermuteTiles;
    const int64_t blocks = groups * 8 * ublocks;
    if (blocks > 0x7fffffffLL) return PVAMD_E_SHAPE;
    hipLaunchKernelGGL(composed_unpermute_kernel, dim3((unsigned)blocks), dim3(kUnpermuteWaves * 64), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(rec), index, P, stride, A, ublocks, out_val, out_grad);
    return (int)hipGetLastError();
}

// One query launch, one un-permute launch.  (Alternating the two over chunks of configurations small enough for the
// packed records to stay in the 256 MB Infinity Cache was slower: README-size C4 1.66 ms unchunked, 1.89 / 2.10 / 3.29
// ms with 192 / 96 / 32 MB chunks -- the query kernel lives off the L2 reuse between MANY configurations of a tile.)
extern "C" int pvamd_composed_query_bucketed(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A,
                                             const float* sorted_points, const int32_t* inv, int64_t P, int64_t Pp,
                                             float* scratch, float* out_val, float* out_grad, int32_t flags,
                                             void* stream) {
    if (P < 1 || Pp < P) return PVAMD_E_SHAPE;
    if (!inv || !out_val || !out_grad) return PVAMD_E_NULL;
    const int rc = pvamd_composed_query_packed(grids, S, tf, A, sorted_points, Pp, scratch, flags, stream);
    if (rc != 0) return rc;
    return pvamd_unpack_records(scratch, inv, P, Pp, A, out_val, out_grad, stream);
}

extern "C" int pvamd_composed_query(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A,
                                    const float* points, int64_t P, float* out_val, float* out_grad,
                                    int32_t* out_leaf, int32_t flags, void* stream) {
    if (S < 1 || A < 1 || P < 0 || S >= kUnnormalised) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grids || !tf || !out_val || !out_grad || !points) return PVAMD_E_NULL;
    if (!aligned_to(grids, 8) || !aligned_to(tf, 4) || !aligned_to(points, 4) || !aligned_to(out_val, 4) ||
        !aligned_to(out_grad, 4))
        return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    // The leaf descriptors live in device memory; whether any of them asks for float64 index arithmetic is not
    // known host-side, so the kernels are built for the general case and test the (wave-uniform) flag per leaf.
    // Two kernels.  The wave-tile kernel (256 points per wave through LDS, per-tile leaf mask, configuration-fastest block
    // order) is the one that scales with configurations and exploits coherent points; the one-point-per-lane kernel has
    // 4x the parallelism, no per-tile overhead, and 44 VGPRs (8 waves per SIMD without spills).  Measured (tools/
    // scalar_probe.py, ms, wave-tile | one-point-per-lane): C3 4M random 0.106 | 0.095, C3 Morton-sorted 0.064 | 0.074,
    // C4 200 x 262k random 0.85 | 0.97, sorted 0.60 | 0.82, README-size grids 4.9 | 6.4 and 1.0 | 2.6.  So: a single
    // configuration, or too few tiles to fill the chip (100k points x 8 leaves: 36 -> 17 us), takes the per-lane kernel.
    // Either kernel takes ANY point count and any 4-byte aligned buffers (round 2 sent P % 4 != 0 -- the reference README's
    // own M = 15,251 -- to the per-lane kernel: the wave-tile kernel's 16-byte stores wanted aligned (A, P) rows; they do
    // not: common.h f32x4_u) and the wave-tile kernel covers a ragged end with a last tile moved back to end at point P - 1:
    // one launch per slab (fewer than 256 points always take the per-lane kernel).
    // (flags bits 1 and 2, for tools/scalar_probe.py and the tests: force the per-lane / the wave-tile kernel.)
    const int64_t ntiles = (P + kTilePoints - 1) / kTilePoints;
    // Round 6: the chunk-grouped kernel with the sort inside the workgroup (composed_query_fused) wherever there are enough
    // (chunk, configuration) workgroups to fill the chip twice over and the grids are L2-resident: scattered points cost it a
    // sixth more work per chunk and save a third of the leaf loop (C3 0.081 -> see profiles/r06_composed_variants.txt); a caller
    // with scratch and several configurations sorts once instead (pvamd_group_points + pvamd_composed_query_grouped).
    {
        const int64_t nchunks = (P + kFusedChunk - 1) / kFusedChunk;
        const bool plain_flags = !(flags & (PVAMD_COMPOSED_INLINE_EXACT | PVAMD_COMPOSED_LEGACY_LEAF_LOOP | PVAMD_COMPOSED_FORCE_PER_LANE |
                                            PVAMD_COMPOSED_FORCE_WAVE_TILE | PVAMD_COMPOSED_POINTS_FASTEST | PVAMD_COMPOSED_NO_GROUPING));
        const bool fused = P >= kFusedChunk && S < kNoLeaf &&
                           ((plain_flags && nchunks * (int64_t)A >= kFusedMinBlocks) || (flags & PVAMD_COMPOSED_FORCE_FUSED));
        if (fused) {
            int64_t cap = ((int64_t)65536 + A - 1) / A;
            if (cap > 65535) cap = 65535;
            const unsigned gy = (unsigned)(nchunks < cap ? nchunks : (cap < 1 ? 1 : cap));
            hipLaunchKernelGGL((composed_query_fused<kFusedWaves, PVAMD_COMPOSED_PPP>), dim3(A, gy), dim3(kFusedWaves * 64), 0, s, grids, S, tf,
                               A, points, nchunks, P, out_val, out_grad, out_leaf, 0);
            return (int)hipGetLastError();
        }
    }
    const bool wave_tiles = P >= kTilePoints && ((A >= 2 && ntiles * (int64_t)A >= kWaveTileMinTiles && !(flags & 2)) || (flags & 4));
    // up to ~65536 blocks in total, split over the A configurations: about one 256-point tile per wave.  (Sweep on C4,
    // 200 x 262,144: 1024 blocks 1.40 ms, 4096 1.17, 8192 1.13, 32768 1.09, 65536 1.08 -- the hardware dispatcher
    // balances better than a grid-stride loop over unequal tiles.)
    // The configuration is a grid dimension: blockIdx.x (any count) in the wave-tile kernel and in the per-lane kernel's
    // default order; blockIdx.y (<= 65535) in the per-lane kernel's points-fastest order, whose larger batches go out in
    // slabs (the kernels take the slab's first configuration and index transforms / outputs with the global one)
    const bool a_is_y = !wave_tiles && (flags & 8);  // only this order carries the configuration in gridDim.y (<= 65535)
    const int slab = a_is_y ? kConfigSlab : (kConfigSlab < 65535 ? kConfigSlab : A);
    for (int a0 = 0; a0 < A; a0 += slab) {
        const int An = A - a0 < slab ? A - a0 : slab;
        int64_t cap = ((int64_t)65536 + An - 1) / An;
        if (cap > 65535) cap = 65535;  // gridDim.y
        if (wave_tiles) {
            const int64_t need = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
            const unsigned gy = (unsigned)(need < cap ? need : (cap < 1 ? 1 : cap));
            if (flags & PVAMD_COMPOSED_INLINE_EXACT)
                hipLaunchKernelGGL((composed_query_wave<PVAMD_COMPOSED_PPP, kInlineExact, false, 0>), dim3(An, gy), dim3(kWavesPerBlock * 64), 0, s,
                                   grids, S, tf, A, points, ntiles, P, out_val, out_grad, out_leaf, a0);
            else if (legacy_leaf_loop(flags, S))
                hipLaunchKernelGGL((composed_query_wave<PVAMD_COMPOSED_PPP, kEstimate, false, 0>), dim3(An, gy), dim3(kWavesPerBlock * 64), 0, s,
                                   grids, S, tf, A, points, ntiles, P, out_val, out_grad, out_leaf, a0);
            else
                hipLaunchKernelGGL((composed_query_wave<PVAMD_COMPOSED_PPP, kEstimate, false, 1>), dim3(An, gy), dim3(kWavesPerBlock * 64),
                                   0, s, grids, S, tf, A, points, ntiles, P, out_val, out_grad, out_leaf, a0);
        } else {
            const int64_t need = (P + 255) / 256;
            const unsigned gx = (unsigned)(need < cap ? need : (cap < 1 ? 1 : cap));
            // configuration fastest unless flag 8 asks for the other order (tuning): with it the A = 200 README case runs
            // 0.076 instead of 0.079 ms on the slice and 0.265 instead of 0.323 ms on random points (21 MB link grids)
            const int cf = (flags & 8) ? 0 : 1;
            // large, gather-bound grids (the inline-exact hint): the two minima cost a second gather of the winner's record
            // and lose 6-12 % there (README-size link grids, 200 x 15,251: 0.087 vs 0.082 ms on the slice, 0.300 vs 0.268 ms
            // on random points); L2-resident grids gain 7-18 % (C4 per-lane 0.975 -> 0.899 ms, README slice 0.062 -> 0.058)
            if (legacy_leaf_loop(flags, S) || (flags & PVAMD_COMPOSED_INLINE_EXACT))
                hipLaunchKernelGGL(composed_query_scalar<false>, cf ? dim3(An, gx) : dim3(gx, An), dim3(256), 0, s, grids, S, tf, A, points,
                                   (int64_t)0, P, out_val, out_grad, out_leaf, a0, cf);
            else
                hipLaunchKernelGGL(composed_query_scalar<true>, cf ? dim3(An, gx) : dim3(gx, An), dim3(256), 0, s, grids, S, tf, A, points,
                                   (int64_t)0, P, out_val, out_grad, out_leaf, a0, cf);
        }
    }
    return (int)hipGetLastError();
}

// ---- chunk-grouped query (round 6): scratch layout = sorted points | run bounds | perm ----
static inline int64_t group_chunks(int64_t P) { return (P + kGroupChunk - 1) / kGroupChunk; }
static inline int64_t group_bounds_offset(int64_t P) { return group_chunks(P) * kGroupChunk * 12; }
static inline int64_t group_perm_offset(int64_t P) { return group_bounds_offset(P) + group_chunks(P) * kGroupWaves * 32; }

extern "C" int64_t pvamd_group_chunk_points(void) { return kGroupChunk; }

extern "C" int64_t pvamd_group_scratch_bytes(int64_t P) {
    if (P < kGroupChunk) return 0;
    return group_perm_offset(P) + group_chunks(P) * kGroupChunk * 2;
}

extern "C" int pvamd_group_points(const float* points, int64_t P, void* scratch, void* stream) {
    if (P < kGroupChunk) return PVAMD_E_SHAPE;
    if (!points || !scratch) return PVAMD_E_NULL;
    if (!aligned_to(points, 4) || !aligned_to(scratch, 16)) return PVAMD_E_ALIGN;
    char* base = static_cast<char*>(scratch);
    hipLaunchKernelGGL((group_points_kernel<kGroupWaves>), dim3((unsigned)group_chunks(P)), dim3(kGroupWaves * 64), 0, (hipStream_t)stream,
                       points, P, reinterpret_cast<float*>(base), reinterpret_cast<float*>(base + group_bounds_offset(P)),
                       reinterpret_cast<uint16_t*>(base + group_perm_offset(P)));
    return (int)hipGetLastError();
}

extern "C" int pvamd_composed_query_grouped(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A, const void* scratch,
                                            int64_t P, float* out_val, float* out_grad, int32_t* out_leaf, int32_t flags,
                                            void* stream) {
    if (S < 1 || A < 1 || P < kGroupChunk || S >= kNoLeaf) return PVAMD_E_SHAPE;
    if (flags & (PVAMD_COMPOSED_INLINE_EXACT | PVAMD_COMPOSED_LEGACY_LEAF_LOOP)) return PVAMD_E_MODE;
    const bool packed = (flags & PVAMD_COMPOSED_OUT_PACKED) != 0;
    if (!grids || !tf || !out_val || (!packed && !out_grad) || !scratch) return PVAMD_E_NULL;
    if (packed && out_grad) return PVAMD_E_MODE;  // records go to out_val alone
    if (!aligned_to(grids, 8) || !aligned_to(tf, 4) || !aligned_to(scratch, 16) || !aligned_to(out_val, packed ? 16 : 4) ||
        !aligned_to(out_grad, 4))
        return PVAMD_E_ALIGN;
    const char* base = static_cast<const char*>(scratch);
    const int64_t nchunks = group_chunks(P);
    // configuration = blockIdx.x (any count), chunks = blockIdx.y (grid-strided beyond 65535)
    int64_t cap = ((int64_t)65536 + A - 1) / A;
    if (cap > 65535) cap = 65535;
    const unsigned gy = (unsigned)(nchunks < cap ? nchunks : (cap < 1 ? 1 : cap));
    const float* sorted = reinterpret_cast<const float*>(base);
    const float* bounds = reinterpret_cast<const float*>(base + group_bounds_offset(P));
    const uint16_t* perm = reinterpret_cast<const uint16_t*>(base + group_perm_offset(P));
    if (packed)
        hipLaunchKernelGGL((composed_query_grouped<kGroupWaves, PVAMD_COMPOSED_PPP, true>), dim3(A, gy), dim3(kGroupWaves * 64), 0,
                           (hipStream_t)stream, grids, S, tf, A, sorted, bounds, perm, nchunks, P, out_val, out_grad, out_leaf, 0);
    else
        hipLaunchKernelGGL((composed_query_grouped<kGroupWaves, PVAMD_COMPOSED_PPP, false>), dim3(A, gy), dim3(kGroupWaves * 64), 0,
                           (hipStream_t)stream, grids, S, tf, A, sorted, bounds, perm, nchunks, P, out_val, out_grad, out_leaf, 0);
    return (int)hipGetLastError();
}

extern "C" int pvamd_composed_query_f64(const pvamd_grid_t* grids, int32_t S, const double* tf, int32_t A,
                                        const double* points, int64_t P, double* out_val, double* out_grad,
                                        int32_t* out_leaf, void* stream) {
    if (S < 1 || A < 1 || P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grids || !tf || !out_val || !out_grad || !points) return PVAMD_E_NULL;
    if (!aligned_to(grids, 8) || !aligned_to(tf, 8) || !aligned_to(points, 8) || !aligned_to(out_val, 8) || !aligned_to(out_grad, 8))
        return PVAMD_E_ALIGN;
    int64_t gy = (P + 255) / 256;
    int64_t cap = ((int64_t)65536 + A - 1) / A;
    if (cap > 65535) cap = 65535;  // gridDim.y; the kernel grid-strides over the points
    if (gy > cap) gy = cap;
    hipLaunchKernelGGL(composed_query_f64_kernel, dim3(A, (unsigned)(gy < 1 ? 1 : gy)), dim3(256), 0, (hipStream_t)stream, grids, S,
                       tf, A, points, P, out_val, out_grad, out_leaf);
    return (int)hipGetLastError();
}

extern "C" int pvamd_transform_points(const float* tf, int32_t A, const float* points, int64_t P, float* out,
                                      void* stream) {
    if (A < 1 || A > 65535 || P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!tf || !points || !out) return PVAMD_E_NULL;
    hipLaunchKernelGGL(transform_points_kernel, dim3(stream_grid(P, 256), A), dim3(256), 0, (hipStream_t)stream, tf, A, points,
                       P, out);
    return (int)hipGetLastError();
}

extern "C" int pvamd_compose_merge(const float* tf, int32_t A, int64_t P, const float* leaf_val, const float* leaf_grad,
                                   int32_t s, int32_t first, float* best_val, float* best_grad, int32_t* best_leaf,
                                   void* stream) {
    if (A < 1 || A > 65535 || P < 0 || s < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!tf || !leaf_val || !leaf_grad || !best_val || !best_grad) return PVAMD_E_NULL;
    hipLaunchKernelGGL(compose_merge_kernel, dim3(stream_grid(P, 256), A), dim3(256), 0, (hipStream_t)stream, tf, A, P,
                       leaf_val, leaf_grad, s, first, best_val, best_grad, best_leaf);
    return (int)hipGetLastError();
}
