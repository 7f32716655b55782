// Z-order (Morton) keys and the order-preserving float <-> uint32 code used for atomic bounds.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "exact_math.h"

namespace pvamd {

// 30-bit Z-order key of a point inside the box [lo, hi]: 3 x 10 bits, interleaved (NaN coordinates -> cell 0).
PVAMD_DEV unsigned morton_key30(float x, float y, float z, const float lo[3], const float hi[3]) {
    const float p[3] = {x, y, z};
    unsigned key = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float t = (p[d] - lo[d]) / fmaxf(hi[d] - lo[d], 1e-30f) * 1023.f;
        t = fminf(fmaxf(t, 0.f), 1023.f);  // NaN -> 0
        unsigned c = (unsigned)t;
        c = (c | (c << 16)) & 0x030000FFu;  // spread 10 bits to every third position
        c = (c | (c << 8)) & 0x0300F00Fu;
        c = (c | (c << 4)) & 0x030C30C3u;
        c = (c | (c << 2)) & 0x09249249u;
        key |= c << d;
    }
    return key;
}

// Floats folded through an order-preserving map to uint32 so that atomicMin / atomicMax give float bounds.
PVAMD_DEV unsigned order_code(float f) {
    const unsigned b = (unsigned)__float_as_int(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
PVAMD_DEV float order_decode(unsigned c) {
    return __int_as_float((int)((c & 0x80000000u) ? (c & 0x7fffffffu) : ~c));
}

}  // namespace pvamd
