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
//   vlo/vhi: the range test "min <= p <= max" is defined in the index dtype; for a float64 range and a float32 p it
//            is equivalent to comparing p with min rounded UP / max rounded DOWN to float32 -- exact, and fp32-only.
//   inv32/err32: index estimate t = (p - fmin32) * inv32 in fp32.  Its distance to the exact quotient is bounded by
//            err32[d] * (|p| + 1) + 2e-7 * |t| (derivation in DESIGN.md, "index fast path"); only when t is closer than
//            that to a half-integer do the kernels redo the reference's exact IEEE division.
#include <cmath>
#include <cfloat>
extern "C" int pvamd_grid_finalize(pvamd_grid_t* g) {
    if (!g) return PVAMD_E_NULL;
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
        g->vlo[d] = flo;
        g->vhi[d] = fhi;
        g->inv32[d] = (float)(1.0 / res);
        // |t32 - exact| <= (|p| + 2|min|) * 2^-24 / res + |t| * 2^-23 (+ float64 round-off, negligible).  The estimate is
        // only ever used for in-range p, where |p| <= max(|lo|, |hi|) and |t| <= shape: one constant per dimension, with
        // a 2x safety margin on the first term
        const double amin = std::fabs(lo), pmax = std::fmax(std::fabs(lo), std::fabs(hi));
        g->err32[d] = (float)((2.0 * 5.97e-8 / res * (1.0 + 2.0 * amin)) * (pmax + 1.0) + 2.5e-7 * (double)g->shape[d]);
    }
    g->finalized = 1;
    return 0;
}
