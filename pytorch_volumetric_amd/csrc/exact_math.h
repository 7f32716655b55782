// Correctly rounded single operations that the compiler may not fuse or approximate.
// HIP's __fadd_rn/__fmul_rn/... are plain `+`/`*` (contractable into fma) and __fsqrt_rn is the *native*
// (approximate) square root unless OCML_BASIC_ROUNDED_OPERATIONS is defined, so the arithmetic contract of this
// library is spelled with these helpers instead.  Division and sqrt rely on hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt; the Makefile also passes -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace pvamd {

#ifndef PVAMD_DEV
#define PVAMD_DEV __device__ __forceinline__
#endif

#pragma clang fp contract(off)
PVAMD_DEV float add_rn(float a, float b) { return a + b; }
PVAMD_DEV float sub_rn(float a, float b) { return a - b; }
PVAMD_DEV float mul_rn(float a, float b) { return a * b; }
PVAMD_DEV float div_rn(float a, float b) { return a / b; }
PVAMD_DEV float sqrt_rn(float a) { return __builtin_sqrtf(a); }
// v_sqrt_f32: 1 ulp, denormal inputs flush to 0 (true root <= 1.09e-19).  Only for conservative bounds that carry
// their own slack -- never for a value that reaches an output.
PVAMD_DEV float fast_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }

}  // namespace pvamd
