// Library identity entry points of the C ABI (include/pvamd.h).
#include "common.h"

extern "C" int pvamd_abi_version(void) { return PVAMD_ABI_VERSION; }

extern "C" const char* pvamd_build_info(void) {
    return "libpvamd gfx950 (MI355X/CDNA4) hipcc " __VERSION__ " built " __DATE__;
}

extern "C" int pvamd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Derived descriptor fields (see include/pvamd.h).  Host-only arithmetic.
//   vlo/vhi: the float32 end points of the interval of valid p.  Default rule: "min <= p <= max" is defined in the index
//            dtype; for a float64 range and a float32 p it is equivalent to comparing p with min rounded UP / max rounded
//            DOWN to float32 -- exact, and fp32-only.  PVAMD_RULE_VALID_ON_INDEX: p is valid when its ROUNDED index lies
//            in [0, shape); the index statement is monotone in p, so the valid set is still an interval and its end
//            points are found by bisection over the float32 ordering with the exact statement itself.
//   inv32/err32: index estimate t = (p - fmin32) * inv32 in fp32.  Its distance to the exact quotient is bounded by
//            err32[d] (derivation in DESIGN.md, "index fast path"); only when t is closer than that to a half-integer do
//            the kernels redo the reference's exact IEEE division -- which is also where the rounding rules differ.
#include <cmath>
#include <cfloat>
#include <cstring>

namespace {

double round_rule(int rule, double q) {
    if (rule & PVAMD_RULE_ROUND_HALF_AWAY) return std::round(q);
    if (rule & PVAMD_RULE_ROUND_FLOOR_HALF) return std::floor(q + 0.5);
    return std::rint(q);
}
float round_rule(int rule, float q) {
    if (rule & PVAMD_RULE_ROUND_HALF_AWAY) return std::roundf(q);
    if (rule & PVAMD_RULE_ROUND_FLOOR_HALF) return std::floor(q + 0.5f);
    return std::rintf(q);
}

// the exact statement of grid_lookup.h voxel_index_1d, validity only
bool valid_1d(const pvamd_grid_t* g, int d, float p) {
    if (g->index_f64) {
        const double pd = (double)p;
        if (!(g->rule & PVAMD_RULE_VALID_ON_INDEX)) return g->dmin[d] <= pd && pd <= g->dmax[d];
        const double kq = round_rule(g->rule, (pd - g->dmin[d]) / g->dres[d]);
        return kq >= 0.0 && kq <= (double)(g->shape[d] - 1);
    }
    if (!(g->rule & PVAMD_RULE_VALID_ON_INDEX)) return g->fmin[d] <= p && p <= g->fmax[d];
    const float kq = round_rule(g->rule, (p - g->fmin[d]) / g->fres[d]);
    return kq >= 0.f && kq <= (float)(g->shape[d] - 1);
}

// float32 <-> a signed integer that orders like the floats (-0 and +0 share 0)
int32_t ordinal(float f) {
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i >= 0 ? i : (int32_t)(0x80000000u - (uint32_t)i);
}
float from_ordinal(int32_t o) {
    const int32_t i = o >= 0 ? o : (int32_t)(0x80000000u - (uint32_t)o);
    float f;
    std::memcpy(&f, &i, 4);
    return f;
}

// the end point of the valid interval on the side `dir` (-1: lowest valid p, +1: highest) given one valid p
float valid_end(const pvamd_grid_t* g, int d, float inside, int dir) {
    const int64_t limit = dir < 0 ? (int64_t)ordinal(-FLT_MAX) : (int64_t)ordinal(FLT_MAX);
    int64_t in = ordinal(inside), out = in, step = 1 << 8;
    for (;;) {  // gallop outwards to an invalid p
        out += dir * step;
        if (dir * (out - limit) >= 0) {
            out = limit;
            if (valid_1d(g, d, from_ordinal((int32_t)out))) return from_ordinal((int32_t)out);  // valid to the end of float32
            break;
        }
        if (!valid_1d(g, d, from_ordinal((int32_t)out))) break;
        in = out;
        step *= 2;
    }
    while (dir * (out - in) > 1) {  // `in` valid, `out` invalid
        const int64_t mid = in + (out - in) / 2;
        if (valid_1d(g, d, from_ordinal((int32_t)mid))) in = mid;
        else out = mid;
    }
    return from_ordinal((int32_t)in);
}

}  // namespace

extern "C" int pvamd_grid_finalize(pvamd_grid_t* g) {
    if (!g) return PVAMD_E_NULL;
    const int known = PVAMD_RULE_VALID_ON_INDEX | PVAMD_RULE_ROUND_HALF_AWAY | PVAMD_RULE_ROUND_FLOOR_HALF | PVAMD_RULE_RES_F64;
    if ((g->rule & ~known) || ((g->rule & PVAMD_RULE_ROUND_HALF_AWAY) && (g->rule & PVAMD_RULE_ROUND_FLOOR_HALF))) return PVAMD_E_MODE;
    for (int d = 0; d < 3; ++d) {
        if (g->shape[d] < 2) return PVAMD_E_SHAPE;
        // a surface bounding box is (min, max) with min <= max: the kernels' median-of-three form of sdf.py:559-567
        // relies on it (NaN bounds, i.e. "no box", pass)
        if (g->bb_min[d] > g->bb_max[d] || g->dbb_min[d] > g->dbb_max[d]) return PVAMD_E_SHAPE;
        const double lo = g->index_f64 ? g->dmin[d] : (double)g->fmin[d];
        const double hi = g->index_f64 ? g->dmax[d] : (double)g->fmax[d];
        const double res = g->index_f64 ? g->dres[d] : (double)g->fres[d];
        if (g->index_f64) {  // the fp32 triple is the rounded float64 one (the estimate's operands)
            g->fmin[d] = (float)g->dmin[d];
            g->fmax[d] = (float)g->dmax[d];
            g->fres[d] = (float)g->dres[d];
        }
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo < lo) flo = std::nextafterf(flo, INFINITY);
        if ((double)fhi > hi) fhi = std::nextafterf(fhi, -INFINITY);
        double reach = 0.0;  // how far beyond [lo, hi] a valid p can lie
        if (g->rule & PVAMD_RULE_VALID_ON_INDEX) {
            const float mid = (float)(0.5 * (lo + hi));
            if (!(res > 0.0) || !valid_1d(g, d, mid)) return PVAMD_E_SHAPE;  // a range that does not hold its own middle
            flo = valid_end(g, d, mid, -1);
            fhi = valid_end(g, d, mid, +1);
            reach = res;
        }
        g->vlo[d] = flo;
        g->vhi[d] = fhi;
        g->inv32[d] = (float)(1.0 / res);
        // |t32 - exact| <= (|p| + 2|min|) * 2^-24 / res + |t| * 2^-23 (+ float64 round-off, negligible).  The estimate is
        // only ever used for valid p, where |p| <= max(|lo|, |hi|) (+ res on the index rule) and |t| <= shape (+ 1): one
        // constant per dimension, with a 2x safety margin on the first term
        const double amin = std::fabs(lo), pmax = std::fmax(std::fabs(lo), std::fabs(hi)) + reach;
        g->err32[d] = (float)((2.0 * 5.97e-8 / res * (1.0 + 2.0 * amin)) * (pmax + 1.0) +
                              2.5e-7 * (double)(g->shape[d] + (reach > 0.0 ? 1 : 0)));
    }
    // one bound for all three axes (the largest): lets the kernels test the three estimates with one compare (grid_lookup.h
    // estimate_unsure); a larger bound only sends a few more points to the exact statements
    const float worst = std::fmax(g->err32[0], std::fmax(g->err32[1], g->err32[2]));
    for (int d = 0; d < 3; ++d) g->err32[d] = worst;
    // range_n2: the kernels' own statements (grid_lookup.h / composed.hip: t = med3(p - bb_min, p - bb_max, 0) per axis, n2 =
    // fma(tz, tz, fma(ty, ty, tx * tx)), every operation a correctly rounded float32 one) are monotone in p on either side of
    // the box, so over the valid interval |t| is largest at vlo or vhi and n2 at the corner that takes the larger one per axis
    float pad[3];
    for (int d = 0; d < 3; ++d) {
        float worst_t = 0.f;
        const float ends[2] = {g->vlo[d], g->vhi[d]};
        for (float p : ends) {
            const float d1 = p - g->bb_min[d], d2 = p - g->bb_max[d];  // d1 >= d2
            float t = d1 < 0.f ? d1 : (d2 > 0.f ? d2 : 0.f);           // median of (d1, d2, 0)
            if (d1 != d1 || d2 != d2) t = NAN;
            worst_t = (t != t || worst_t != worst_t) ? NAN : std::fmax(worst_t, std::fabs(t));
        }
        pad[d] = worst_t;
    }
    const float n2 = std::fmaf(pad[2], pad[2], std::fmaf(pad[1], pad[1], pad[0] * pad[0]));
    g->range_n2 = (n2 != n2) ? INFINITY : n2;
    g->reserved0 = 0.f;
    g->finalized = 1;
    return 0;
}
