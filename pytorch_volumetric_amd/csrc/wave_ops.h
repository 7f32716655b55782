// Wave64 reductions on the DPP datapath.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace pvamd {

#ifndef PVAMD_DEV
#define PVAMD_DEV __device__ __forceinline__
#endif

// wave64 min over lanes: v_min_f32 with a DPP source (butterfly within rows of 16, then row broadcasts); the result is
// read from lane 63.  One vector instruction per step; the s_nop covers the VALU-write -> DPP-read hazard, which the
// compiler does not track through inline assembly.  NaN inputs are ignored (v_min_f32 returns the other operand).
#define PVAMD_DPP_MIN(v, ctrl) asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " ctrl : "+v"(v))
PVAMD_DEV float wave_min(float v) {
    PVAMD_DPP_MIN(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "row_mirror row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    PVAMD_DPP_MIN(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
PVAMD_DEV float wave_max(float v) { return -wave_min(-v); }

// the same within each run of 8 consecutive lanes: every lane of the run gets the run's min / max
PVAMD_DEV float group8_min(float v) {
    PVAMD_DPP_MIN(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
    return v;
}
PVAMD_DEV float group8_max(float v) { return -group8_min(-v); }

// ... and within each row of 16 consecutive lanes
PVAMD_DEV float group16_min(float v) {
    PVAMD_DPP_MIN(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
    PVAMD_DPP_MIN(v, "row_mirror row_mask:0xf bank_mask:0xf");
    return v;
}
PVAMD_DEV float group16_max(float v) { return -group16_min(-v); }

}  // namespace pvamd
