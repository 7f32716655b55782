// Device-side voxel index arithmetic and nearest-voxel lookup shared by the cached / composed / chamfer kernels.
// gfx950 only.  Restates (not copies) what the reference asks of its value-range view:
//   index  = round_half_even((p - min) / res) as integer      (sdf.py:537, TorchMultidimView.ensure_index_key)
//   flat   = (kx*ny + ky)*nz + kz                              (sdf.py:538, ravel_multi_index)
//   valid  = all_d(min_d <= p_d <= max_d)                      (sdf.py:540, get_valid_values)
// in float32 or float64 according to pvamd_grid_t::index_f64 (see include/pvamd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "exact_math.h"
#include "../../include/pvamd.h"

namespace pvamd {

// hipcc contracts a*b+c into fma by default; every expression whose rounding is part of the contract below is
// written with explicit fmaf / __f*_rn so that the CPU oracle can state the same sequence.
#define PVAMD_DEV __device__ __forceinline__

// One packed (val, gx, gy, gz) record.  The cache always lives in device (global) memory, but a pointer read out of a
// descriptor that itself sits in memory is a generic one to the compiler, which then emits flat_load (aperture check, and the
// load counts against lgkmcnt as well as vmcnt): say which address space it is.
typedef float record_f32x4 __attribute__((ext_vector_type(4)));
typedef const record_f32x4 __attribute__((address_space(1))) * global_record_ptr;
PVAMD_DEV float4 load_record(const float* vox, int flat) {
    const record_f32x4 r = ((global_record_ptr)(uintptr_t)vox)[flat];
    return make_float4(r.x, r.y, r.z, r.w);
}

// The quotient -> index step and the validity test under a NON-default pvamd_grid_t::rule (include/pvamd.h): inline, but
// behind ONE wave-uniform branch per coordinate (voxel_index_1d), so that descriptors with the default rule never execute
// them (a noinline call was worse: the by-value descriptor went through scratch).
template <typename T>
PVAMD_DEV T round_by_rule(int rule, T q) {
    if (rule & PVAMD_RULE_ROUND_HALF_AWAY) {
        if constexpr (sizeof(T) == 8) return __builtin_round(q);
        else return __builtin_roundf(q);
    }
    if (rule & PVAMD_RULE_ROUND_FLOOR_HALF) {
        if constexpr (sizeof(T) == 8) return __builtin_floor(q + 0.5);  // the sum rounds in the index dtype (-ffp-contract=off)
        else return __builtin_floorf(add_rn(q, 0.5f));
    }
    if constexpr (sizeof(T) == 8) return __builtin_rint(q);
    else return __builtin_rintf(q);
}

// kq = the rounded quotient as a float: "valid on the index" tests it (0 <= kq <= shape - 1; a NaN / infinite quotient
// fails), "valid on the value" tests p.
PVAMD_DEV bool voxel_index_1d_ruled(const pvamd_grid_t& g, int d, float p, bool f64, long long& k) {
    double kq;
    if (f64) kq = round_by_rule<double>(g.rule, ((double)p - g.dmin[d]) / g.dres[d]);
    else kq = (double)round_by_rule<float>(g.rule, div_rn(sub_rn(p, g.fmin[d]), g.fres[d]));
    k = (long long)kq;
    if (g.rule & PVAMD_RULE_VALID_ON_INDEX) return (kq >= 0.0) && (kq <= (double)(g.shape[d] - 1));
    return f64 ? (g.dmin[d] <= (double)p) && ((double)p <= g.dmax[d]) : (g.fmin[d] <= p) && (p <= g.fmax[d]);
}

// index (as the reference would hold it, int64) and validity of one coordinate: the default statements inline, any other
// rule behind one wave-uniform branch
template <bool F64>
PVAMD_DEV bool voxel_index_1d(const pvamd_grid_t& g, int d, float p, long long& k) {
    if (__builtin_expect(g.rule != 0, 0)) return voxel_index_1d_ruled(g, d, p, F64, k);
    if constexpr (F64) {
        const double pd = (double)p;
        const bool valid = (g.dmin[d] <= pd) && (pd <= g.dmax[d]);
        k = (long long)__builtin_rint((pd - g.dmin[d]) / g.dres[d]);
        return valid;
    } else {
        const bool valid = (g.fmin[d] <= p) && (p <= g.fmax[d]);
        k = (long long)__builtin_rintf(div_rn(sub_rn(p, g.fmin[d]), g.fres[d]));
        return valid;
    }
}

// Full key (as the reference would hold it, int64, possibly outside [0, shape)) + validity.
template <bool F64>
PVAMD_DEV bool voxel_key(const pvamd_grid_t& g, float x, float y, float z, long long key[3]) {
    const bool vx = voxel_index_1d<F64>(g, 0, x, key[0]);
    const bool vy = voxel_index_1d<F64>(g, 1, y, key[1]);
    const bool vz = voxel_index_1d<F64>(g, 2, z, key[2]);
    return vx & vy & vz;
}

// In-bounds flat index for the gather.  Valid points always land in [0, shape) by construction
// ((max-min)/res rounds to shape-1); the clamp only protects the load against a malformed descriptor.
template <bool F64>
PVAMD_DEV bool voxel_flat(const pvamd_grid_t& g, float x, float y, float z, int& flat) {
    long long key[3];
    const bool valid = voxel_key<F64>(g, x, y, z, key);
    int kx = (int)key[0], ky = (int)key[1], kz = (int)key[2];
    kx = min(max(kx, 0), g.shape[0] - 1);
    ky = min(max(ky, 0), g.shape[1] - 1);
    kz = min(max(kz, 0), g.shape[2] - 1);
    flat = (kx * g.shape[1] + ky) * g.shape[2] + kz;
    return valid;
}

// ---- fast path used by the query kernels (bit-identical to the exact statements above) ----
// Range test in fp32: the set of valid float32 p is an interval per coordinate whatever the rule (the index statement is
// monotone in p); pvamd_grid_finalize() finds its float32 end points vlo / vhi -- for the default rule "min <= p <= max"
// in the index dtype with the bounds rounded inward to float32.
PVAMD_DEV bool in_range(const pvamd_grid_t& g, float x, float y, float z) {
    return (g.vlo[0] <= x) & (x <= g.vhi[0]) & (g.vlo[1] <= y) & (y <= g.vhi[1]) & (g.vlo[2] <= z) & (z <= g.vhi[2]);
}

// The same test as a 64-bit LANE MASK, for the instruction-bound composed kernels: the ballot of ONE compare is that
// compare's own SGPR result, and the six masks are combined with scalar ANDs -- whereas the ballot of an AND of compares
// makes the backend materialise the bool (v_cndmask 0 / 1) and compare it with 0 again, two vector instructions per visit.
// Round 5: per axis ONE v_med3_f32 + ONE compare -- x in [lo, hi] <=> med3(x, lo, hi) == x (a NaN fails the ==; -0 == +0 as
// with <=) -- the same six vector instructions as six compares, but three lane masks to AND instead of six (the scalar unit is
// as busy as the vector ones in these kernels): per-lane kernel 0.896 -> 0.866 ms on C4's workload, C3 0.0897 -> 0.0886 ms,
// README slice 0.0572 -> 0.0561 ms; wave-tile kernel within noise (profiles/r05_composed_variants.txt).
PVAMD_DEV uint64_t in_range_mask(const pvamd_grid_t& g, float x, float y, float z) {
#ifndef PVAMD_RANGE_SIX_COMPARES
    return __builtin_amdgcn_ballot_w64(__builtin_amdgcn_fmed3f(x, g.vlo[0], g.vhi[0]) == x) &
           __builtin_amdgcn_ballot_w64(__builtin_amdgcn_fmed3f(y, g.vlo[1], g.vhi[1]) == y) &
           __builtin_amdgcn_ballot_w64(__builtin_amdgcn_fmed3f(z, g.vlo[2], g.vhi[2]) == z);
#endif
    return __builtin_amdgcn_ballot_w64(g.vlo[0] <= x) & __builtin_amdgcn_ballot_w64(x <= g.vhi[0]) &
           __builtin_amdgcn_ballot_w64(g.vlo[1] <= y) & __builtin_amdgcn_ballot_w64(y <= g.vhi[1]) &
           __builtin_amdgcn_ballot_w64(g.vlo[2] <= z) & __builtin_amdgcn_ballot_w64(z <= g.vhi[2]);
}

// Index of an in-range coordinate: multiply-first estimate in fp32; unless it lies within its rounding bound of a
// half-integer (where the estimate and the reference's quotient could round differently) it IS the reference index.
// Otherwise -- a few points per million -- the exact IEEE division of voxel_index_1d is redone.
template <bool F64>
PVAMD_DEV int voxel_index_fast(const pvamd_grid_t& g, int d, float p) {
    const float t = mul_rn(sub_rn(p, g.fmin[d]), g.inv32[d]);
    const float kc = __builtin_rintf(t);
    const float half_dist = sub_rn(0.5f, fabsf(sub_rn(t, kc)));
    if (half_dist > g.err32[d]) return (int)kc;  // NaN-safe: any NaN fails the comparison and takes the exact path
    long long k;
    voxel_index_1d<F64>(g, d, p, k);
    return (int)k;
}

// Wave votes without the 0/1 round trip through a VGPR that __all / __any cost: the compare mask is already an SGPR pair.
PVAMD_DEV bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }
PVAMD_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

// Correctly rounded sqrt of a sum of squares (n2 >= 0 or NaN).  v_sqrt_f32 is within 1 ulp; the neighbour whose
// residual says so replaces it (the same two-sided test the compiler's expansion of sqrtf uses).  That test needs
// n2 >= 2^-96 to keep its residuals normal -- the compiler pre-scales smaller inputs; here they (rare: a distance below
// 3.5e-15) take the generic path through one wave-uniform branch.  0, inf and NaN fall through the test unchanged.
PVAMD_DEV float sqrt_rn_sumsq(float n2) {
    const bool tiny = (unsigned)(__float_as_int(n2) - 1) < (unsigned)(0x0F800000 - 1);  // 0 < n2 < 2^-96
    if (__builtin_expect(wave_any(tiny), 0)) return sqrt_rn(n2);
    float s = __builtin_amdgcn_sqrtf(n2);
    const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
    const float r_dn = fmaf(-s_dn, s, n2), r_up = fmaf(-s_up, s, n2);
    s = (r_dn <= 0.f) ? s_dn : s;
    s = (r_up > 0.f) ? s_up : s;
    return s;
}

// Whether some coordinate's estimate t sits within its rounding bound of a half-integer (off[d] = t - rint(t)): the estimate
// and the reference's quotient could then round differently.  pvamd_grid_finalize() stores ONE bound (the largest of the three
// axes') in every err32[d], so the three tests "0.5 - |off_d| > err" are one: 0.5 - max_d |off_d| > err (the subtraction is
// monotone) -- v_max3_f32 with |.| modifiers, a subtract and a compare instead of three of each.  A NaN offset is ignored by
// the max: it only arises from a non-finite coordinate, which is never in range, and whose index nobody uses.
PVAMD_DEV bool estimate_unsure(const pvamd_grid_t& g, const float off[3]) {
#ifdef PVAMD_UNSURE_PER_AXIS
    return !(sub_rn(0.5f, fabsf(off[0])) > g.err32[0]) | !(sub_rn(0.5f, fabsf(off[1])) > g.err32[1]) | !(sub_rn(0.5f, fabsf(off[2])) > g.err32[2]);
#else
    const float worst = __builtin_fmaxf(__builtin_fmaxf(fabsf(off[0]), fabsf(off[1])), fabsf(off[2]));
    return !(sub_rn(0.5f, worst) > g.err32[0]);
#endif
}

// Index estimate only (no exact fallback inline): returns the flat index of the estimate and sets `unsure` when some
// coordinate sits within its rounding bound of a half-integer -- the caller then redoes that POINT with the exact
// statements (voxel_flat<F64>) outside its hot loop, which keeps the float64 division sequence (and its registers) out
// of it.  The gather is kept in bounds by clamping the flat index once (a no-op for a well-formed descriptor, whose
// in-range points always index inside the grid).
PVAMD_DEV int voxel_flat_estimate(const pvamd_grid_t& g, float x, float y, float z, bool& unsure) {
    const float p[3] = {x, y, z};
    int k[3];
    float off[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float t = mul_rn(sub_rn(p[d], g.fmin[d]), g.inv32[d]);
        const float kc = __builtin_rintf(t);
        off[d] = sub_rn(t, kc);
        k[d] = (int)kc;
    }
    unsure |= estimate_unsure(g, off);
    const unsigned flat = (unsigned)((k[0] * g.shape[1] + k[1]) * g.shape[2] + k[2]);
    const unsigned last = (unsigned)(g.shape[0] * g.shape[1] * g.shape[2] - 1);
    return (int)(flat < last ? flat : last);
}

// The estimate with the exact statements inline, behind ONE rare branch for all three coordinates: the flagged lanes redo
// them with the IEEE division of the leaf's index dtype (a loop, not unrolled: the float64 division sequence exists
// once).  For grids where the flag is common (large coordinate / resolution ratios, e.g. 21 MB README-size link grids:
// 22 % of the wave passes hold a flagged (lane, leaf)) this is cheaper than redoing whole points after the loop.
PVAMD_DEV int voxel_flat_in_range_fused(const pvamd_grid_t& g, float x, float y, float z) {
    const float p[3] = {x, y, z};
    int k[3];
    float off[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float t = mul_rn(sub_rn(p[d], g.fmin[d]), g.inv32[d]);
        const float kc = __builtin_rintf(t);
        off[d] = sub_rn(t, kc);
        k[d] = (int)kc;
    }
    const bool unsure = estimate_unsure(g, off);
    if (__builtin_expect(wave_any(unsure), 0)) {
        if (unsure) {
#pragma unroll 1
            for (int d = 0; d < 3; ++d) {
                long long kd;
                if (g.index_f64) voxel_index_1d<true>(g, d, p[d], kd);
                else voxel_index_1d<false>(g, d, p[d], kd);
                k[d] = (int)kd;
            }
        }
    }
    const unsigned flat = (unsigned)((k[0] * g.shape[1] + k[1]) * g.shape[2] + k[2]);
    const unsigned last = (unsigned)(g.shape[0] * g.shape[1] * g.shape[2] - 1);
    return (int)(flat < last ? flat : last);
}

template <bool F64>
PVAMD_DEV int voxel_flat_in_range(const pvamd_grid_t& g, float x, float y, float z) {
    int kx = voxel_index_fast<F64>(g, 0, x), ky = voxel_index_fast<F64>(g, 1, y), kz = voxel_index_fast<F64>(g, 2, z);
    kx = min(max(kx, 0), g.shape[0] - 1);  // no-ops for a well-formed descriptor; keep the gather in bounds regardless
    ky = min(max(ky, 0), g.shape[1] - 1);
    kz = min(max(kz, 0), g.shape[2] - 1);
    return (kx * g.shape[1] + ky) * g.shape[2] + kz;
}

// BOUNDING_BOX fallback (sdf.py:559-571): per component dmin = max(bbmin - p, 0), dmax = max(p - bbmax, 0),
// t = dmin + dmax, negated where dmin > 0; val = |t|, grad = t / |t|  (0/0 = NaN when the point is inside
// the box, exactly as the reference).  Returns |t| and leaves the UNNORMALISED vector in t[] so that callers that
// only need the gradient of a winning leaf can defer the three divisions.
PVAMD_DEV float bounding_box_vector(const pvamd_grid_t& g, float x, float y, float z, float t[3]) {
    const float p[3] = {x, y, z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float lo = sub_rn(g.bb_min[d], p[d]);
        const bool lo_active = lo > 0.f;
        lo = lo_active ? lo : 0.f;
        float hi = sub_rn(p[d], g.bb_max[d]);
        hi = (hi > 0.f) ? hi : 0.f;
        const float s = add_rn(lo, hi);
        t[d] = lo_active ? -s : s;
    }
    return sqrt_rn(fmaf(t[2], t[2], fmaf(t[1], t[1], mul_rn(t[0], t[0]))));
}

PVAMD_DEV float4 bounding_box_sdf(const pvamd_grid_t& g, float x, float y, float z) {
    float t[3];
    const float n = bounding_box_vector(g, x, y, z, t);
    return make_float4(n, div_rn(t[0], n), div_rn(t[1], n), div_rn(t[2], n));
}

// (val, gx, gy, gz) for one point in the leaf frame; `valid` reports the range test.
// Round 5: the statements the composed kernels had already been brought down to -- range test as three med3 + compare, ONE
// rare branch for the exact index statements of all three axes (voxel_flat_in_range_fused), the bounding-box vector as one
// v_med3_f32 per component (equal to sdf.py:559-567's max / add / negate for every input incl. NaN and infinities: negation and
// adding 0 are exact), its norm by v_sqrt_f32 + the two-sided residual test.  The 1M-point C2 launch is as much
// instruction-bound as memory-bound (profiles/r05_cq_geometry.txt): ~110 -> ~75 vector instructions per mixed point.
// STREAMING = the statements of rounds 1-4, kept for launches far beyond the Infinity Cache (> 8M points), which are bound by
// the L1 -> L2 request path and run SLOWER with the shorter look-up (64M points 332 -> 353 us, every point gathering 406 -> 532:
// the gathers of a tile crowd the TCP's pending-request slots sooner; profiles/r05_cq_geometry.txt).
template <bool F64, bool STREAMING = false>
PVAMD_DEV float4 cached_lookup(const pvamd_grid_t& g, float x, float y, float z, bool& valid) {
#ifdef PVAMD_CQ_OLD_LOOKUP
    constexpr bool kOld = true;
#else
    constexpr bool kOld = STREAMING;
#endif
    if constexpr (kOld) {
    valid = in_range(g, x, y, z);
    if (valid) {
        // (g is a kernarg here: the compiler already knows vox is global, and routing it through load_record's integer cast
        // changes the schedule of cached_query_wave for the worse: 64M points 0.626 -> 0.578 of 8 TB/s)
        return reinterpret_cast<const float4*>(g.vox)[voxel_flat_in_range<F64>(g, x, y, z)];
    }
    if (g.oob_mode == PVAMD_OOB_BOUNDING_BOX) {
        return bounding_box_sdf(g, x, y, z);
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);  // LOOKUP_GT_SDF: zeros (sdf.py:546-547), caller fills in
    } else {
    valid = (__builtin_amdgcn_fmed3f(x, g.vlo[0], g.vhi[0]) == x) & (__builtin_amdgcn_fmed3f(y, g.vlo[1], g.vhi[1]) == y) &
            (__builtin_amdgcn_fmed3f(z, g.vlo[2], g.vhi[2]) == z);
    if (valid) return reinterpret_cast<const float4*>(g.vox)[voxel_flat_in_range_fused(g, x, y, z)];
    if (g.oob_mode == PVAMD_OOB_BOUNDING_BOX) {
        const float ta = __builtin_amdgcn_fmed3f(sub_rn(x, g.bb_min[0]), sub_rn(x, g.bb_max[0]), 0.f);
        const float tb = __builtin_amdgcn_fmed3f(sub_rn(y, g.bb_min[1]), sub_rn(y, g.bb_max[1]), 0.f);
        const float tc = __builtin_amdgcn_fmed3f(sub_rn(z, g.bb_min[2]), sub_rn(z, g.bb_max[2]), 0.f);
        const float n = sqrt_rn_sumsq(fmaf(tc, tc, fmaf(tb, tb, mul_rn(ta, ta))));  // sdf.py:568
        return make_float4(n, div_rn(ta, n), div_rn(tb, n), div_rn(tc, n));          // sdf.py:570
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);  // LOOKUP_GT_SDF: zeros (sdf.py:546-547), caller fills in
    }
}

// ---- float64 query points (sdf.py:545-547: output dtype = query dtype; torch promotion makes the index arithmetic,
// the range test and the BOUNDING_BOX branch float64) ----
PVAMD_DEV bool voxel_key_f64(const pvamd_grid_t& g, const double p[3], long long key[3]) {
    bool valid = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double kq = round_by_rule<double>(g.rule, (p[d] - g.dmin[d]) / g.dres[d]);
        key[d] = (long long)kq;
        if (__builtin_expect(g.rule & PVAMD_RULE_VALID_ON_INDEX, 0)) valid &= (kq >= 0.0) && (kq <= (double)(g.shape[d] - 1));
        else valid &= (g.dmin[d] <= p[d]) && (p[d] <= g.dmax[d]);
    }
    return valid;
}

PVAMD_DEV int clamped_flat(const pvamd_grid_t& g, const long long key[3]) {
    const int kx = min(max((int)key[0], 0), g.shape[0] - 1);
    const int ky = min(max((int)key[1], 0), g.shape[1] - 1);
    const int kz = min(max((int)key[2], 0), g.shape[2] - 1);
    return (kx * g.shape[1] + ky) * g.shape[2] + kz;
}

// (val, gx, gy, gz) of one float64 point: the cached record widened exactly, or the bounding-box statements in float64
PVAMD_DEV bool cached_lookup_f64(const pvamd_grid_t& g, const double p[3], double& v, double& gx, double& gy, double& gz) {
    long long key[3];
    const bool valid = voxel_key_f64(g, p, key);
    v = gx = gy = gz = 0.0;
    if (valid) {
        const float4 r = load_record(g.vox, clamped_flat(g, key));
        v = (double)r.x; gx = (double)r.y; gy = (double)r.z; gz = (double)r.w;
    } else if (g.oob_mode == PVAMD_OOB_BOUNDING_BOX) {
        double t[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double lo = g.dbb_min[d] - p[d];
            const bool lo_active = lo > 0.0;
            lo = lo_active ? lo : 0.0;
            double hi = p[d] - g.dbb_max[d];
            hi = (hi > 0.0) ? hi : 0.0;
            const double s = lo + hi;
            t[d] = lo_active ? -s : s;
        }
        v = __builtin_sqrt(__builtin_fma(t[2], t[2], __builtin_fma(t[1], t[1], t[0] * t[0])));
        gx = t[0] / v; gy = t[1] / v; gz = t[2] / v;
    }
    return valid;
}

// x' = M p for a row-major 4x4 (column-vector convention), k-ordered fma chain -- the rounding sequence of an
// f32 MFMA / a bmm k-loop: ((m0*px (+) m1*py) (+) m2*pz) + m3.
PVAMD_DEV float affine_row(float m0, float m1, float m2, float m3, float px, float py, float pz) {
    return add_rn(fmaf(m2, pz, fmaf(m1, py, mul_rn(m0, px))), m3);
}

}  // namespace pvamd
