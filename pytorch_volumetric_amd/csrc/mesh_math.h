// Point-vs-triangle device math for the mesh kernels.  gfx950 only.
//   closest point on a triangle: Ericson, "Real-Time Collision Detection" 5.1.5 -- the routine Embree's point-query
//     tutorial ships and open3d's RaycastingScene::ComputeClosestPoints runs (reference call site sdf.py:134)
//   ray hit: Embree's Moeller-Trumbore triangle test with tnear = 0, tfar = inf -- what open3d's
//     count_intersections runs for a [origin, direction] ray (reference call site sdf.py:153)
// All products/sums whose rounding matters are spelled with explicit fmaf / __f*_rn (hipcc would otherwise contract
// freely); DESIGN.md "Arithmetic contract" lists the sequences.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "exact_math.h"

namespace pvamd {

#ifndef PVAMD_DEV
#define PVAMD_DEV __device__ __forceinline__
#endif

struct V3 {
    float x, y, z;
};

PVAMD_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
PVAMD_DEV V3 sub(V3 a, V3 b) { return V3{sub_rn(a.x, b.x), sub_rn(a.y, b.y), sub_rn(a.z, b.z)}; }
PVAMD_DEV float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, mul_rn(a.x, b.x))); }
PVAMD_DEV V3 cross(V3 a, V3 b) {
    return V3{fmaf(a.y, b.z, -mul_rn(a.z, b.y)), fmaf(a.z, b.x, -mul_rn(a.x, b.z)),
              fmaf(a.x, b.y, -mul_rn(a.y, b.x))};
}
PVAMD_DEV V3 madd(float s, V3 d, V3 o) { return V3{fmaf(s, d.x, o.x), fmaf(s, d.y, o.y), fmaf(s, d.z, o.z)}; }

// Closest point of triangle (a,b,c) to p.  Region tests in Ericson's order: A, B, C, AB, AC, BC, interior.
PVAMD_DEV V3 closest_point_triangle(V3 p, V3 a, V3 b, V3 c) {
    const V3 ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) return a;
    const V3 bp = sub(p, b);
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) return b;
    const V3 cp = sub(p, c);
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) return c;
    const float vc = fmaf(d1, d4, -mul_rn(d3, d2));
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        return madd(div_rn(d1, sub_rn(d1, d3)), ab, a);
    }
    const float vb = fmaf(d5, d2, -mul_rn(d1, d6));
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        return madd(div_rn(d2, sub_rn(d2, d6)), ac, a);
    }
    const float va = fmaf(d3, d6, -mul_rn(d5, d4));
    const float d43 = sub_rn(d4, d3), d56 = sub_rn(d5, d6);
    if (va <= 0.f && d43 >= 0.f && d56 >= 0.f) {
        return madd(div_rn(d43, add_rn(d43, d56)), sub(c, b), b);
    }
    const float denom = div_rn(1.f, add_rn(add_rn(va, vb), vc));
    const float v = mul_rn(vb, denom), w = mul_rn(vc, denom);
    return madd(w, ac, madd(v, ab, a));
}

// 1 if the ray org + t*dir, t > 0, crosses triangle (v0,v1,v2); edges inclusive.
PVAMD_DEV int ray_hits_triangle(V3 org, V3 dir, V3 v0, V3 v1, V3 v2) {
    const V3 e1 = sub(v0, v1), e2 = sub(v2, v0);
    const V3 Ng = cross(e2, e1);
    const V3 C = sub(v0, org);
    const V3 R = cross(C, dir);
    const float den = dot(Ng, dir);
    const float absden = fabsf(den);
    const float sgn = den < 0.f ? -1.f : 1.f;
    const float U = mul_rn(dot(R, e2), sgn);
    const float V = mul_rn(dot(R, e1), sgn);
    const float T = mul_rn(dot(Ng, C), sgn);
    const bool hit = (den != 0.f) && (U >= 0.f) && (V >= 0.f) && (add_rn(U, V) <= absden) && (T > 0.f);
    return hit ? 1 : 0;
}

PVAMD_DEV uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Irwin-Hall(12) unit-variance variate from twelve 16-bit uniforms: exactly representable, identical on host and
// device.  Stands in for the reference's unseeded np.random.randn (sdf.py:149).
PVAMD_DEV float jitter_normal(uint64_t seed, int64_t index, int c) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)index * 9u + (uint64_t)(c * 3 + k)));
        s += (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)((h >> 48) & 0xFFFF);
    }
    return mul_rn((float)(s - 393210), 1.0f / 65536.0f);
}

// ray direction = float32(bounding_box(padding=1.0).max + 1e-4 * N(0,1)), the sum taken in float64 (sdf.py:147-150)
PVAMD_DEV V3 jitter_dir(const double ray_dir[3], uint64_t seed, int64_t index) {
    return V3{(float)__fma_rn(1e-4, (double)jitter_normal(seed, index, 0), ray_dir[0]),
              (float)__fma_rn(1e-4, (double)jitter_normal(seed, index, 1), ray_dir[1]),
              (float)__fma_rn(1e-4, (double)jitter_normal(seed, index, 2), ray_dir[2])};
}

}  // namespace pvamd
