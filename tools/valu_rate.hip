// VALU issue-rate table for gfx950 (not part of the product): how many cycles one SIMD needs per wave64 instruction of
// each opcode the composed / mesh kernels are made of.  This is the denominator of a "VALU-bound" claim: the kernels'
// SQ_INSTS_VALU mix priced with these numbers gives the time the vector ALUs alone need.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/valu_rate.hip -o tools/valu_rate.bin && ./tools/valu_rate.bin
// Method: every wave runs ITERS x 32 instructions over 8 independent register chains (no dependent-issue stalls) and
// reads s_memtime before/after; W waves per SIMD are resident (W = 1, 2, 4, 8).  cycles/instruction/SIMD =
// elapsed shader cycles x 1 / (ITERS x 32 x W), the wall-clock figure from HIP events is printed beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY4(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL_F32(NAME, ASM)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc) {                                        \
        float a[8], b = out[threadIdx.x & 7], c = out[8 + (threadIdx.x & 7)];                                        \
        for (int i = 0; i < 8; ++i) a[i] = out[16 + i] + threadIdx.x;                                                \
        asm volatile("s_mov_b32 s22, 0x55555555\n s_mov_b32 s23, 0x55555555" ::: "s22", "s23");                      \
        const long long t0 = __builtin_readcyclecounter();                                                           \
        for (int it = 0; it < ITERS; ++it) {                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23");     \
            }                                                                                                        \
        }                                                                                                            \
        const long long t1 = __builtin_readcyclecounter();                                                           \
        float s = 0;                                                                                                 \
        for (int i = 0; i < 8; ++i) s += a[i];                                                                       \
        if (s == 12345.f) out[0] = s;                                                                                \
        if ((threadIdx.x & 63) == 0) { const long long w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * w] = t0; cyc[2 * w + 1] = t1; }                   \
    }

#define KERNEL_PK(NAME, ASM)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc) {                                        \
        f32x2 a[8], b = {out[threadIdx.x & 7], out[1]}, c = {out[8 + (threadIdx.x & 7)], out[2]};                    \
        for (int i = 0; i < 8; ++i) a[i] = f32x2{out[16 + i] + threadIdx.x, out[17 + i]};                            \
        const long long t0 = __builtin_readcyclecounter();                                                           \
        for (int it = 0; it < ITERS; ++it) {                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21");     \
            }                                                                                                        \
        }                                                                                                            \
        const long long t1 = __builtin_readcyclecounter();                                                           \
        float s = 0;                                                                                                 \
        for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;                                                            \
        if (s == 12345.f) out[0] = s;                                                                                \
        if ((threadIdx.x & 63) == 0) { const long long w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * w] = t0; cyc[2 * w + 1] = t1; }                   \
    }

#define KERNEL_F64(NAME, ASM)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc) {                                        \
        double a[8], b = out[threadIdx.x & 7], c = out[8 + (threadIdx.x & 7)];                                       \
        for (int i = 0; i < 8; ++i) a[i] = out[16 + i] + threadIdx.x;                                                \
        const long long t0 = __builtin_readcyclecounter();                                                           \
        for (int it = 0; it < ITERS; ++it) {                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21");     \
            }                                                                                                        \
        }                                                                                                            \
        const long long t1 = __builtin_readcyclecounter();                                                           \
        double s = 0;                                                                                                \
        for (int i = 0; i < 8; ++i) s += a[i];                                                                       \
        if (s == 12345.0) out[0] = (float)s;                                                                         \
        if ((threadIdx.x & 63) == 0) { const long long w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * w] = t0; cyc[2 * w + 1] = t1; }                   \
    }

KERNEL_F32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
KERNEL_F32(k_add_f32, "v_add_f32 %0, %1, %0")
KERNEL_F32(k_mul_f32, "v_mul_f32 %0, %1, %0")
KERNEL_F32(k_max_f32, "v_max_f32 %0, %1, %0")
KERNEL_F32(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_F32(k_cmp_lt_f32, "v_cmp_lt_f32 vcc, %1, %0")
KERNEL_F32(k_cmp_e64, "v_cmp_lt_f32 s[20:21], %1, %0")
// v_cndmask_b32 with its select mask in an SGPR pair that nothing in the loop rewrites -- the form the composed / mesh kernels
// issue (inverse_ballot of a lane mask).  Round 4's rows used the implicit VCC and read 9.5 ns at every occupancy: that is a VCC
// read-after-(clobber) chain of the harness, not an issue rate -- the shipped composed loop issues a v_cndmask every few
// instructions at 1.35 ns per instruction overall.  Those rows are gone.
KERNEL_F32(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %1, %2, s[22:23]")
KERNEL_F32(k_cndmask_sgpr_rw, "v_cndmask_b32_e64 %0, %0, %1, s[22:23]")
KERNEL_F32(k_med3_f32, "v_med3_f32 %0, %1, %2, %0")
KERNEL_F32(k_min3_f32, "v_min3_f32 %0, %1, %2, %0")
// the composed loop's pattern: a compare that writes an SGPR pair next to arithmetic, and compare -> select -> fma
KERNEL_F32(k_cmp_fma, "v_cmp_lt_f32 s[20:21], %1, %0\n v_fma_f32 %0, %1, %2, %0")                                                   // 2 / slot
KERNEL_F32(k_cmp_cnd_fma, "v_cmp_lt_f32 s[20:21], %1, %0\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_fma_f32 %0, %1, %2, %0")      // 3 / slot
KERNEL_F32(k_cmp_med3_fma, "v_cmp_le_f32 s[20:21], %1, %0\n v_med3_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0")                  // 3 / slot
KERNEL_F32(k_rndne_f32, "v_rndne_f32 %0, %0")
KERNEL_F32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL_F32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_F32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F32(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
KERNEL_F32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL_F32(k_add_u32, "v_add_u32 %0, %1, %0")
KERNEL_F32(k_lshl_add_u32, "v_lshl_add_u32 %0, %1, 2, %0")
KERNEL_F32(k_max_i32, "v_max_i32 %0, %1, %0")
KERNEL_F32(k_div_scale_f32, "v_div_scale_f32 %0, vcc, %1, %2, %0")
KERNEL_F32(k_div_fmas_f32, "v_div_fmas_f32 %0, %1, %2, %0")
KERNEL_F32(k_div_fixup_f32, "v_div_fixup_f32 %0, %1, %2, %0")
KERNEL_F32(k_fma_mix, "v_fma_f32 %0, %1, %2, %0\n v_cndmask_b32 %0, %0, %2, vcc")  // 2 instructions per slot
KERNEL_F32(k_sub_max, "v_sub_f32 %0, %1, %0\n v_max_f32 %0, %0, %2")
KERNEL_PK(k_pk_fma_f32, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_PK(k_pk_add_f32, "v_pk_add_f32 %0, %1, %0")
KERNEL_PK(k_pk_mul_f32, "v_pk_mul_f32 %0, %1, %0")
KERNEL_PK(k_lshl_add_u64, "v_lshl_add_u64 %0, %1, 2, %0")
KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNEL_F64(k_add_f64, "v_add_f64 %0, %1, %0")
KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %1, %0")
KERNEL_F64(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL_F64(k_rcp_f64, "v_rcp_f64 %0, %0")

typedef void (*kern_t)(float*, long long*);
struct Entry { const char* name; kern_t k; int per_slot; };

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %.0f MHz\n", prop.name, cus, prop.clockRate / 1000.0);
    float* out;
    long long* cyc;
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(out, 0, 4096));
    const int maxWaves = cus * 4 * 8;
    CK(hipMalloc(&cyc, sizeof(long long) * 2 * maxWaves));
    std::vector<Entry> es = {
        {"v_fma_f32", k_fma_f32, 1}, {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1}, {"v_max_f32", k_max_f32, 1},
        {"v_mov_b32", k_mov_b32, 1}, {"v_cmp_lt_f32 (vcc)", k_cmp_lt_f32, 1}, {"v_cmp_lt_f32 (sgpr pair)", k_cmp_e64, 1},
        {"v_cndmask_b32 (sgpr-pair mask)", k_cndmask_sgpr, 1}, {"v_cndmask_b32 (sgpr mask, dst=src0)", k_cndmask_sgpr_rw, 1},
        {"v_med3_f32", k_med3_f32, 1}, {"v_min3_f32", k_min3_f32, 1}, {"v_cmp(sgpr)+v_fma (2/slot)", k_cmp_fma, 2},
        {"v_cmp+v_cndmask+v_fma (3/slot)", k_cmp_cnd_fma, 3}, {"v_cmp+v_med3+v_fma (3/slot)", k_cmp_med3_fma, 3},
        {"v_rndne_f32", k_rndne_f32, 1}, {"v_cvt_i32_f32", k_cvt_i32_f32, 1},
        {"v_sqrt_f32", k_sqrt_f32, 1}, {"v_rcp_f32", k_rcp_f32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1},
        {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_add_u32", k_add_u32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1},
        {"v_max_i32", k_max_i32, 1}, {"v_div_scale_f32", k_div_scale_f32, 1}, {"v_div_fmas_f32", k_div_fmas_f32, 1},
        {"v_div_fixup_f32", k_div_fixup_f32, 1}, {"v_fma_f32+v_cndmask (2/slot)", k_fma_mix, 2},
        {"v_sub_f32+v_max_f32 (2/slot)", k_sub_max, 2}, {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1}, {"v_pk_mul_f32", k_pk_mul_f32, 1},
        {"v_lshl_add_u64", k_lshl_add_u64, 1},
        {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
        {"v_rndne_f64", k_rndne_f64, 1}, {"v_rcp_f64", k_rcp_f64, 1},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("per opcode and W resident waves per SIMD: ns = wall-clock (HIP events) per wave64 instruction per SIMD; own = s_memtime ticks\n"
           "between two instructions of ONE wave (median); conc = sum of per-wave busy ticks / (span x SIMDs) = waves actually co-resident per SIMD;\n"
           "GHz = s_memtime ticks per wall ns\n");
    printf("%-36s |", "opcode");
    for (int W : {1, 2, 4, 6, 8}) printf("  W=%d: ns    own  conc   GHz |", W);
    printf("\n");
    std::vector<long long> h(2 * maxWaves);
    for (auto& e : es) {
        printf("%-36s |", e.name);
        for (int W : {1, 2, 4, 6, 8}) {
            const int blocks = cus * W;  // 256-thread blocks: 4 waves = one per SIMD; W blocks per CU
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const int waves = blocks * 4;
            CK(hipMemcpy(h.data(), cyc, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost));
            std::vector<double> own(waves);
            long long lo = h[0], hi = h[1];
            double busy = 0;
            for (int w = 0; w < waves; ++w) {
                own[w] = (double)(h[2 * w + 1] - h[2 * w]);
                busy += own[w];
                lo = std::min(lo, h[2 * w]);
                hi = std::max(hi, h[2 * w + 1]);
            }
            std::sort(own.begin(), own.end());
            const double instr = (double)ITERS * 32 * e.per_slot;
            printf(" %6.3f %6.2f %5.2f %5.2f |", ms * 1e6 / (instr * W), own[waves / 2] / instr, busy / ((double)(hi - lo) * cus * 4),
                   (double)(hi - lo) / (ms * 1e6));
        }
        printf("\n");
    }
    return 0;
}
