// Host-side helpers shared by the C-ABI translation units.  gfx950 (MI355X) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pvamd.h"

// A/B builds (tools/build_variant.sh compiles ONE translation unit with extra -D knobs and -DPVAMD_VARIANT="name: flags") carry
// their name in the library: _lib.load() refuses a library that exports pvamd_variant unless PVAMD_ALLOW_VARIANT=1 is set, so a
// tuning build cannot be picked up by the product path or the tests by accident.  The product build never defines it.
#ifdef PVAMD_VARIANT
extern "C" __attribute__((visibility("default"), weak)) const char* pvamd_variant(void) { return PVAMD_VARIANT; }
#endif

namespace pvamd {

// native clang vectors: the non-temporal builtins and 16-B global_load/store_dwordx4 want these, not HIP's structs
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// same 16-byte vector, but allowed to alias plain float storage (LDS slices accessed both as floats and as float4s)
typedef f32x4 __attribute__((may_alias)) f32x4_alias;
// the same 16 bytes at an address that is only 4-byte aligned: gfx950 global_load/store_dwordx4 take any dword address
// (amdhsa runs with unaligned access mode on), so an (A, P) output row that starts at a * P floats with P odd still
// leaves as 16-byte stores -- the compiler emits the same instruction as for f32x4
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

// Wave-synchronous LDS exchange: lanes of ONE wave hand data to each other through LDS without a block barrier.  The
// hardware executes a wave's DS instructions in order, but the compiler reasons per thread and will move a lane's
// LDS store past loads that (for that lane) cannot alias -- seen in the ISA of the first version of cached_query_wave.
// A wavefront-scope release/acquire fence pair around a wave_barrier pins the order; it emits no s_barrier.
#define PVAMD_WAVE_SYNC()                                         \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

constexpr int kNumCU = 256;         // MI355X: 8 XCDs x 32 CUs
constexpr int kMaxBlocksPerCU = 8;  // memory-bound kernels: cap the grid at 256 CU x 8 and grid-stride the rest

static inline bool aligned_to(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// grid size for a streaming kernel over n items with `block` threads
static inline unsigned stream_grid(int64_t n, int block) {
    const int64_t need = (n + block - 1) / block;
    const int64_t cap = (int64_t)kNumCU * kMaxBlocksPerCU;
    return (unsigned)(need < cap ? (need < 1 ? 1 : need) : cap);
}

static inline int check_grid(const pvamd_grid_t& g, bool need_vox = true) {
    if (need_vox && !g.vox) return PVAMD_E_NULL;
    if (need_vox && !aligned_to(g.vox, 16)) return PVAMD_E_ALIGN;
    for (int d = 0; d < 3; ++d) {
        if (g.shape[d] < 2) return PVAMD_E_SHAPE;
    }
    if ((int64_t)g.shape[0] * g.shape[1] * g.shape[2] > (int64_t)INT32_MAX) return PVAMD_E_SHAPE;
    if (g.oob_mode != PVAMD_OOB_LOOKUP_GT_SDF && g.oob_mode != PVAMD_OOB_BOUNDING_BOX) return PVAMD_E_MODE;
    if (!g.finalized) return PVAMD_E_MODE;  // pvamd_grid_finalize() was not called
    return 0;
}

}  // namespace pvamd
