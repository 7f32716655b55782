// Kernel-variant microbenchmark for the CachedSDF query (not part of the product): A/B variants of the C2 kernel in
// one process, interleaved rounds, HIP-event timing.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. tools/kbench.hip -o /tmp/kbench && /tmp/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include "../pytorch_volumetric_amd/csrc/common.h"
#include "../pytorch_volumetric_amd/csrc/grid_lookup.h"

using namespace pvamd;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// ---- V_COPY_STRIDED: same global access pattern as the product kernel, no compute ----
__global__ __launch_bounds__(256) void copy_strided(const f32x4* __restrict__ pts4, int64_t ng, f32x4* __restrict__ val4, f32x4* __restrict__ grad4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += stride) {
        const f32x4 a = __builtin_nontemporal_load(pts4 + 3 * i), b = __builtin_nontemporal_load(pts4 + 3 * i + 1), c = __builtin_nontemporal_load(pts4 + 3 * i + 2);
        __builtin_nontemporal_store(f32x4{a.x, a.w, b.z, c.y}, val4 + i);
        __builtin_nontemporal_store(a, grad4 + 3 * i);
        __builtin_nontemporal_store(b, grad4 + 3 * i + 1);
        __builtin_nontemporal_store(c, grad4 + 3 * i + 2);
    }
}

// ---- V_COPY_LINEAR: 12 B/pt read + 16 B/pt written, every instruction a contiguous 1 KB per wave ----
__global__ __launch_bounds__(256) void copy_linear(const f32x4* __restrict__ in, int64_t n_in, f32x4* __restrict__ out, int64_t n_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t; i < n_in; i += stride) {
        const f32x4 a = __builtin_nontemporal_load(in + i);
        __builtin_nontemporal_store(a, out + i);
    }
    for (int64_t i = n_in + t; i < n_out; i += stride) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, out + i);
}

// ---- product kernel shape (strided vec4), templated on index dtype / gather on-off / nontemporal ----
template <bool F64, bool GATHER, bool NT>
__global__ __launch_bounds__(256) void q_strided(const pvamd_grid_t g, const f32x4* __restrict__ pts4, int64_t ng, f32x4* __restrict__ val4, f32x4* __restrict__ grad4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += stride) {
        f32x4 a, b, c;
        if (NT) { a = __builtin_nontemporal_load(pts4 + 3 * i); b = __builtin_nontemporal_load(pts4 + 3 * i + 1); c = __builtin_nontemporal_load(pts4 + 3 * i + 2); }
        else { a = pts4[3 * i]; b = pts4[3 * i + 1]; c = pts4[3 * i + 2]; }
        bool v0, v1, v2, v3;
        float4 r0, r1, r2, r3;
        if (GATHER) {
            r0 = cached_lookup<F64>(g, a.x, a.y, a.z, v0); r1 = cached_lookup<F64>(g, a.w, b.x, b.y, v1);
            r2 = cached_lookup<F64>(g, b.z, b.w, c.x, v2); r3 = cached_lookup<F64>(g, c.y, c.z, c.w, v3);
        } else {
            int f0, f1, f2, f3;
            v0 = voxel_flat<F64>(g, a.x, a.y, a.z, f0); v1 = voxel_flat<F64>(g, a.w, b.x, b.y, f1);
            v2 = voxel_flat<F64>(g, b.z, b.w, c.x, f2); v3 = voxel_flat<F64>(g, c.y, c.z, c.w, f3);
            r0 = v0 ? make_float4(f0, 0, 0, 0) : bounding_box_sdf(g, a.x, a.y, a.z);
            r1 = v1 ? make_float4(f1, 0, 0, 0) : bounding_box_sdf(g, a.w, b.x, b.y);
            r2 = v2 ? make_float4(f2, 0, 0, 0) : bounding_box_sdf(g, b.z, b.w, c.x);
            r3 = v3 ? make_float4(f3, 0, 0, 0) : bounding_box_sdf(g, c.y, c.z, c.w);
        }
        const f32x4 o0 = {r0.x, r1.x, r2.x, r3.x}, o1 = {r0.y, r0.z, r0.w, r1.y}, o2 = {r1.z, r1.w, r2.y, r2.z}, o3 = {r2.w, r3.y, r3.z, r3.w};
        if (NT) { __builtin_nontemporal_store(o0, val4 + i); __builtin_nontemporal_store(o1, grad4 + 3 * i); __builtin_nontemporal_store(o2, grad4 + 3 * i + 1); __builtin_nontemporal_store(o3, grad4 + 3 * i + 2); }
        else { val4[i] = o0; grad4[3 * i] = o1; grad4[3 * i + 1] = o2; grad4[3 * i + 2] = o3; }
    }
}

// ---- LDS-transposed IO: each wave moves contiguous 1 KB pieces; one point per lane per pass ----
// block = 256 threads handles 1024 points per iteration: loads 3 x (256 lanes x 16 B) contiguous = 12 KB into LDS,
// each thread then reads its 4 points as 12 consecutive floats ... (conflict pattern: stride 12 dwords)
template <bool F64>
__global__ __launch_bounds__(256) void q_lds(const pvamd_grid_t g, const f32x4* __restrict__ pts4, int64_t ntiles, f32x4* __restrict__ val4, f32x4* __restrict__ grad4) {
    __shared__ f32x4 sp[768];   // 1024 points * 12 B
    __shared__ f32x4 sg[768];   // 1024 grads * 12 B
    const int t = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const f32x4* src = pts4 + tile * 768;
        sp[t] = __builtin_nontemporal_load(src + t);
        sp[t + 256] = __builtin_nontemporal_load(src + t + 256);
        sp[t + 512] = __builtin_nontemporal_load(src + t + 512);
        __syncthreads();
        const float* spf = reinterpret_cast<const float*>(sp);
        float* sgf = reinterpret_cast<float*>(sg);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = t + 256 * k;  // point index within the tile: lane-consecutive -> LDS stride 3 dwords (conflict-free)
            bool valid;
            const float4 r = cached_lookup<F64>(g, spf[3 * p], spf[3 * p + 1], spf[3 * p + 2], valid);
            v[k] = r.x;
            sgf[3 * p] = r.y; sgf[3 * p + 1] = r.z; sgf[3 * p + 2] = r.w;
        }
        // values: point p = t + 256k -> val[tile*1024 + p]: 4 scalar-coalesced dword stores (256 B per wave each)
        float* vout = reinterpret_cast<float*>(val4) + tile * 1024;
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], vout + t + 256 * k);
        __syncthreads();
        f32x4* dst = grad4 + tile * 768;
        __builtin_nontemporal_store(sg[t], dst + t);
        __builtin_nontemporal_store(sg[t + 256], dst + t + 256);
        __builtin_nontemporal_store(sg[t + 512], dst + t + 512);
        __syncthreads();
    }
}


// ---- wave-private LDS tiles: no block barrier.  One wave = 256 points per pass: 3 contiguous 1 KB loads, LDS
// transpose to one point per lane (stride-3 dword reads: conflict-free), 4 lookups, LDS transpose back, 4 contiguous
// 1 KB stores.  GMODE: 0 plain gather, 1 nontemporal gather ----
template <bool F64, int GMODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void q_wave(const pvamd_grid_t g, const f32x4* __restrict__ pts4, int64_t ntiles, f32x4* __restrict__ val4, f32x4* __restrict__ grad4) {
    __shared__ __attribute__((aligned(16))) float lds[WAVES][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* spf = lds[wave];
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(spf);
    float* svf = spf + 768;
    const int64_t wstride = (int64_t)gridDim.x * WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += wstride) {
        const f32x4* src = pts4 + tile * 192;
        const f32x4 a = __builtin_nontemporal_load(src + lane), b = __builtin_nontemporal_load(src + lane + 64), c = __builtin_nontemporal_load(src + lane + 128);
        sp[lane] = a; sp[lane + 64] = b; sp[lane + 128] = c;
        PVAMD_WAVE_SYNC();
        float px[4], py[4], pz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int p = lane + 64 * k; px[k] = spf[3 * p]; py[k] = spf[3 * p + 1]; pz[k] = spf[3 * p + 2]; }
        PVAMD_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = lane + 64 * k;
            int flat; float4 r;
            const bool valid = voxel_flat<F64>(g, px[k], py[k], pz[k], flat);
            if (valid) {
                const f32x4* vp = reinterpret_cast<const f32x4*>(g.vox) + flat;
                const f32x4 q = GMODE == 1 ? __builtin_nontemporal_load(vp) : *vp;
                r = make_float4(q.x, q.y, q.z, q.w);
            } else r = bounding_box_sdf(g, px[k], py[k], pz[k]);
            svf[p] = r.x; spf[3 * p] = r.y; spf[3 * p + 1] = r.z; spf[3 * p + 2] = r.w;
        }
        PVAMD_WAVE_SYNC();
        __builtin_nontemporal_store(sp[192 + lane], val4 + tile * 64 + lane);
        f32x4* dst = grad4 + tile * 192;
        __builtin_nontemporal_store(sp[lane], dst + lane);
        __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
        __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128);
        PVAMD_WAVE_SYNC();
    }
}

// ---- dwordx3 per point: lane-consecutive points, every VMEM instruction contiguous (768 B loads/stores, 256 B val
// stores), no LDS.  UNR points per lane in flight. ----
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef f32x3 __attribute__((aligned(4))) f32x3_u;  // 12 bytes of payload at 4-byte alignment
template <bool F64, int UNR>
__global__ __launch_bounds__(256) void q_x3(const pvamd_grid_t g, const float* __restrict__ pts, int64_t P, float* __restrict__ val, float* __restrict__ grad) {
    const int64_t chunk = (int64_t)blockDim.x * UNR;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < P; base += (int64_t)gridDim.x * chunk) {
        float px[UNR], py[UNR], pz[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int64_t i = base + threadIdx.x + (int64_t)k * blockDim.x;
            const f32x3 q = __builtin_nontemporal_load(reinterpret_cast<const f32x3_u*>(pts + 3 * (i < P ? i : P - 1)));
            px[k] = q.x; py[k] = q.y; pz[k] = q.z;
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int64_t i = base + threadIdx.x + (int64_t)k * blockDim.x;
            bool valid;
            const float4 r = cached_lookup<F64>(g, px[k], py[k], pz[k], valid);
            if (i < P) {
                __builtin_nontemporal_store(r.x, val + i);
                const f32x3 o = {r.y, r.z, r.w};
                __builtin_nontemporal_store(o, reinterpret_cast<f32x3_u*>(grad + 3 * i));
            }
        }
    }
}

// ---- q_wave with the next tile's global loads issued before the current tile is processed ----
template <bool F64, int WAVES, int LDP = 0, int STP = 0>
__global__ __launch_bounds__(WAVES * 64) void q_wave_pf(const pvamd_grid_t g, const f32x4* __restrict__ pts4, int64_t ntiles, f32x4* __restrict__ val4, f32x4* __restrict__ grad4) {
    __shared__ __attribute__((aligned(16))) float lds[WAVES][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* spf = lds[wave];
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(spf);
    float* svf = spf + 768;
    const int64_t wstride = (int64_t)gridDim.x * WAVES;
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    f32x4 a, b, c;
    if (tile < ntiles) { const f32x4* src = pts4 + tile * 192; if (LDP) { a = src[lane]; b = src[lane + 64]; c = src[lane + 128]; } else { a = __builtin_nontemporal_load(src + lane); b = __builtin_nontemporal_load(src + lane + 64); c = __builtin_nontemporal_load(src + lane + 128); } }
    for (; tile < ntiles; tile += wstride) {
        sp[lane] = a; sp[lane + 64] = b; sp[lane + 128] = c;
        const int64_t nt = tile + wstride;
        if (nt < ntiles) { const f32x4* src = pts4 + nt * 192; if (LDP) { a = src[lane]; b = src[lane + 64]; c = src[lane + 128]; } else { a = __builtin_nontemporal_load(src + lane); b = __builtin_nontemporal_load(src + lane + 64); c = __builtin_nontemporal_load(src + lane + 128); } }
        PVAMD_WAVE_SYNC();
        float px[4], py[4], pz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int p = lane + 64 * k; px[k] = spf[3 * p]; py[k] = spf[3 * p + 1]; pz[k] = spf[3 * p + 2]; }
        PVAMD_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = lane + 64 * k;
            bool valid;
            const float4 r = cached_lookup<F64>(g, px[k], py[k], pz[k], valid);
            svf[p] = r.x; spf[3 * p] = r.y; spf[3 * p + 1] = r.z; spf[3 * p + 2] = r.w;
        }
        PVAMD_WAVE_SYNC();
        f32x4* dst = grad4 + tile * 192;
        if (STP) { val4[tile * 64 + lane] = sp[192 + lane]; dst[lane] = sp[lane]; dst[lane + 64] = sp[lane + 64]; dst[lane + 128] = sp[lane + 128]; }
        else {
        __builtin_nontemporal_store(sp[192 + lane], val4 + tile * 64 + lane);
        __builtin_nontemporal_store(sp[lane], dst + lane);
        __builtin_nontemporal_store(sp[lane + 64], dst + lane + 64);
        __builtin_nontemporal_store(sp[lane + 128], dst + lane + 128); }
        PVAMD_WAVE_SYNC();
    }
}

__global__ __launch_bounds__(256) void copy16_plain(const f32x4* __restrict__ in, f32x4* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy16_nt(const f32x4* __restrict__ in, f32x4* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

// ---- 64-point sub-tiles, software-pipelined inside each wave: one point per lane per pass, the next sub-tile's loads in
// flight while the current one is looked up.  Meant for SMALL batches (1M points), where a wave otherwise owns a single
// 256-point tile and all waves march through load -> gather -> store in lockstep. ----
template <bool F64, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void q_wave64(const pvamd_grid_t g, const f32x4* __restrict__ pts4, int64_t nsub, f32x4* __restrict__ val4, f32x4* __restrict__ grad4) {
    __shared__ __attribute__((aligned(16))) float lds[WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* spf = lds[wave];
    f32x4_alias* sp = reinterpret_cast<f32x4_alias*>(spf);
    const int64_t wstride = (int64_t)gridDim.x * WAVES;
    int64_t sub = (int64_t)blockIdx.x * WAVES + wave;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (sub < nsub && lane < 48) a = pts4[sub * 48 + lane];
    for (; sub < nsub; sub += wstride) {
        if (lane < 48) sp[lane] = a;
        const int64_t nx = sub + wstride;
        if (nx < nsub && lane < 48) a = pts4[nx * 48 + lane];
        PVAMD_WAVE_SYNC();
        const float px = spf[3 * lane], py = spf[3 * lane + 1], pz = spf[3 * lane + 2];
        PVAMD_WAVE_SYNC();
        bool valid;
        const float4 r = cached_lookup<F64>(g, px, py, pz, valid);
        spf[192 + lane] = r.x; spf[3 * lane] = r.y; spf[3 * lane + 1] = r.z; spf[3 * lane + 2] = r.w;
        PVAMD_WAVE_SYNC();
        if (lane < 16) __builtin_nontemporal_store(sp[48 + lane], val4 + sub * 16 + lane);
        if (lane < 48) __builtin_nontemporal_store(sp[lane], grad4 + sub * 48 + lane);
        PVAMD_WAVE_SYNC();
    }
}

static float frand() { return (float)rand() / (float)RAND_MAX; }

int main(int argc, char** argv) {
    const int64_t P = argc > 1 ? atoll(argv[1]) : (1ll << 26);
    const float margin = argc > 2 ? atof(argv[2]) : 0.05f;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const int burst = argc > 4 ? atoi(argv[4]) : 1;  // launches per timed event pair (back-to-back, like a graph replay)
    // drill-like grid 37x33x40
    pvamd_grid_t g; memset(&g, 0, sizeof(g));
    const double lo[3] = {-0.167981, -0.141332, -0.103716}, res = 0.01; const int shape[3] = {37, 33, 40};
    for (int d = 0; d < 3; ++d) {
        g.shape[d] = shape[d]; g.dmin[d] = lo[d]; g.dmax[d] = lo[d] + res * (shape[d] - 1); g.dres[d] = (g.dmax[d] - g.dmin[d]) / (shape[d] - 1);
        g.fmin[d] = (float)g.dmin[d]; g.fmax[d] = (float)g.dmax[d]; g.fres[d] = (g.fmax[d] - g.fmin[d]) / (float)(shape[d] - 1);
        g.bb_min[d] = (float)(lo[d] + 0.1); g.bb_max[d] = (float)(g.dmax[d] - 0.1);
    }
    g.oob_mode = 1;
    const int64_t nvox = (int64_t)shape[0] * shape[1] * shape[2];
    std::vector<float> hv(nvox * 4); for (auto& x : hv) x = frand();
    float* dvox; CK(hipMalloc(&dvox, nvox * 16)); CK(hipMemcpy(dvox, hv.data(), nvox * 16, hipMemcpyHostToDevice)); g.vox = dvox;
    std::vector<float> hp((size_t)P * 3);
    for (int64_t i = 0; i < P; ++i) for (int d = 0; d < 3; ++d) hp[3 * i + d] = (float)(g.dmin[d] - margin + frand() * (g.dmax[d] - g.dmin[d] + 2 * margin));
    float *dp, *dval, *dgrad; CK(hipMalloc(&dp, P * 12)); CK(hipMalloc(&dval, P * 4)); CK(hipMalloc(&dgrad, P * 12));
    CK(hipMemcpy(dp, hp.data(), P * 12, hipMemcpyHostToDevice));
    float* dout16; CK(hipMalloc(&dout16, P * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int64_t ng = P / 4; const int64_t ntiles = P / 1024; const int64_t nwt = P / 256;
    const unsigned grid = (unsigned)std::min<int64_t>((ng + 255) / 256, 2048);
    const unsigned grid_lds = (unsigned)std::min<int64_t>(ntiles, 2048);
    const f32x4* p4 = (const f32x4*)dp; f32x4* v4 = (f32x4*)dval; f32x4* g4 = (f32x4*)dgrad;
    struct V { const char* name; int id; };
    std::vector<V> vs = {{"copy_linear(28B/pt)", 0}, {"copy_strided", 1}, {"q_strided f64 gather nt (product)", 2}, {"q_strided f32 gather nt", 3},
                         {"q_strided f64 NOgather nt", 4}, {"q_strided f32 NOgather nt", 5}, {"q_strided f64 gather plain-ld/st", 6}, {"q_lds f64", 7}, {"q_lds f32", 8}, {"q_wave f64 plain W4", 9}, {"q_wave f64 ntgather W4", 10}, {"q_wave f32 plain W4", 11}, {"q_wave f64 plain W8", 12}, {"q_wave f64 plain W16", 13}, {"q_wave f64 plain W4 grid4096", 14}, {"q_x3 f64 unr4", 15}, {"q_x3 f64 unr2", 16}, {"q_x3 f64 unr1", 17}, {"q_x3 f64 unr8", 18}, {"q_wave_pf f64 W4 g2048", 19}, {"q_wave_pf f64 W4 g1024", 20}, {"q_x3 f32 unr4", 21}, {"PRODUCT libpvamd.so pvamd_cached_query", 22}, {"q_wave_pf f64 plainLD ntST", 23}, {"q_wave_pf f64 ntLD plainST", 24}, {"q_wave_pf f64 plainLD plainST", 25}, {"copy16 plain (32B/elt: GB/s x 32/28)", 26}, {"copy16 nt", 27}, {"q_wave64 f64 W4 g1024", 28}, {"q_wave64 f64 W4 g2048", 29}, {"q_wave64 f64 W4 g4096", 30}, {"q_wave64 f64 W8 g1024", 31}};
    typedef int (*cq_t)(const pvamd_grid_t*, const float*, int64_t, float*, float*, uint8_t*, void*);
    void* so = dlopen("pytorch_volumetric_amd/csrc/libpvamd.so", RTLD_NOW);
    cq_t product = so ? (cq_t)dlsym(so, "pvamd_cached_query") : nullptr;
    typedef int (*fin_t)(pvamd_grid_t*);
    fin_t finalize = so ? (fin_t)dlsym(so, "pvamd_grid_finalize") : nullptr;
    g.index_f64 = 1;  // the README flow (numpy ranges): float64 index semantics
    if (finalize) finalize(&g); else printf("no pvamd_grid_finalize: variants that need the derived fields will misbehave\n");
    if (!product) printf("libpvamd.so not found (run from the repo root): product row is a no-op\n");
    std::vector<std::vector<float>> times(vs.size());
    auto launch = [&](int id) {
        switch (id) {
            case 0: hipLaunchKernelGGL(copy_linear, dim3(2048), dim3(256), 0, 0, p4, P * 3 / 4, g4, P * 4 / 4); break;  // reads 12B/pt, writes 16 B/pt into grad+val region (grad buf is 12B/pt: write P*3/4 copied + rest fill)
            case 1: hipLaunchKernelGGL(copy_strided, dim3(grid), dim3(256), 0, 0, p4, ng, v4, g4); break;
            case 2: hipLaunchKernelGGL((q_strided<true, true, true>), dim3(grid), dim3(256), 0, 0, g, p4, ng, v4, g4); break;
            case 3: hipLaunchKernelGGL((q_strided<false, true, true>), dim3(grid), dim3(256), 0, 0, g, p4, ng, v4, g4); break;
            case 4: hipLaunchKernelGGL((q_strided<true, false, true>), dim3(grid), dim3(256), 0, 0, g, p4, ng, v4, g4); break;
            case 5: hipLaunchKernelGGL((q_strided<false, false, true>), dim3(grid), dim3(256), 0, 0, g, p4, ng, v4, g4); break;
            case 6: hipLaunchKernelGGL((q_strided<true, true, false>), dim3(grid), dim3(256), 0, 0, g, p4, ng, v4, g4); break;
            case 7: hipLaunchKernelGGL((q_lds<true>), dim3(grid_lds), dim3(256), 0, 0, g, p4, ntiles, v4, g4); break;
            case 8: hipLaunchKernelGGL((q_lds<false>), dim3(grid_lds), dim3(256), 0, 0, g, p4, ntiles, v4, g4); break;
            case 9: hipLaunchKernelGGL((q_wave<true, 0, 4>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 2048)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 10: hipLaunchKernelGGL((q_wave<true, 1, 4>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 2048)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 11: hipLaunchKernelGGL((q_wave<false, 0, 4>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 2048)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 12: hipLaunchKernelGGL((q_wave<true, 0, 8>), dim3((unsigned)std::min<int64_t>((nwt + 7) / 8, 1024)), dim3(512), 0, 0, g, p4, nwt, v4, g4); break;
            case 13: hipLaunchKernelGGL((q_wave<true, 0, 16>), dim3((unsigned)std::min<int64_t>((nwt + 15) / 16, 512)), dim3(1024), 0, 0, g, p4, nwt, v4, g4); break;
            case 15: hipLaunchKernelGGL((q_x3<true, 4>), dim3((unsigned)std::min<int64_t>((P + 1023) / 1024, 4096)), dim3(256), 0, 0, g, dp, P, dval, dgrad); break;
            case 16: hipLaunchKernelGGL((q_x3<true, 2>), dim3((unsigned)std::min<int64_t>((P + 511) / 512, 4096)), dim3(256), 0, 0, g, dp, P, dval, dgrad); break;
            case 17: hipLaunchKernelGGL((q_x3<true, 1>), dim3((unsigned)std::min<int64_t>((P + 255) / 256, 8192)), dim3(256), 0, 0, g, dp, P, dval, dgrad); break;
            case 18: hipLaunchKernelGGL((q_x3<true, 8>), dim3((unsigned)std::min<int64_t>((P + 2047) / 2048, 4096)), dim3(256), 0, 0, g, dp, P, dval, dgrad); break;
            case 19: hipLaunchKernelGGL((q_wave_pf<true, 4>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 2048)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 20: hipLaunchKernelGGL((q_wave_pf<true, 4>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 1024)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 21: hipLaunchKernelGGL((q_x3<false, 4>), dim3((unsigned)std::min<int64_t>((P + 1023) / 1024, 4096)), dim3(256), 0, 0, g, dp, P, dval, dgrad); break;
            case 23: hipLaunchKernelGGL((q_wave_pf<true, 4, 1, 0>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 1024)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 24: hipLaunchKernelGGL((q_wave_pf<true, 4, 0, 1>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 1024)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 25: hipLaunchKernelGGL((q_wave_pf<true, 4, 1, 1>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 1024)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
            case 26: hipLaunchKernelGGL(copy16_plain, dim3(2048), dim3(256), 0, 0, p4, (f32x4*)dout16, P * 3 / 4); break;
            case 27: hipLaunchKernelGGL(copy16_nt, dim3(2048), dim3(256), 0, 0, p4, (f32x4*)dout16, P * 3 / 4); break;
            case 28: hipLaunchKernelGGL((q_wave64<true, 4>), dim3((unsigned)std::min<int64_t>((P / 64 + 3) / 4, 1024)), dim3(256), 0, 0, g, p4, P / 64, v4, g4); break;
            case 29: hipLaunchKernelGGL((q_wave64<true, 4>), dim3((unsigned)std::min<int64_t>((P / 64 + 3) / 4, 2048)), dim3(256), 0, 0, g, p4, P / 64, v4, g4); break;
            case 30: hipLaunchKernelGGL((q_wave64<true, 4>), dim3((unsigned)std::min<int64_t>((P / 64 + 3) / 4, 4096)), dim3(256), 0, 0, g, p4, P / 64, v4, g4); break;
            case 31: hipLaunchKernelGGL((q_wave64<true, 8>), dim3((unsigned)std::min<int64_t>((P / 64 + 7) / 8, 1024)), dim3(512), 0, 0, g, p4, P / 64, v4, g4); break;
            case 22: if (product) product(&g, dp, P, dval, dgrad, nullptr, nullptr); break;
            case 14: hipLaunchKernelGGL((q_wave<true, 0, 4>), dim3((unsigned)std::min<int64_t>((nwt + 3) / 4, 4096)), dim3(256), 0, 0, g, p4, nwt, v4, g4); break;
        }
    };
    // copy_linear writes 16 B/pt: needs an output of P*16 bytes -> reuse a dedicated buffer
    auto launch_fixed = [&](int id) { if (id == 0) hipLaunchKernelGGL(copy_linear, dim3(2048), dim3(256), 0, 0, p4, P * 3 / 4, (f32x4*)dout16, P); else launch(id); };
    for (size_t k = 0; k < vs.size(); ++k) launch_fixed(vs[k].id);
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (size_t k = 0; k < vs.size(); ++k) {
            CK(hipEventRecord(e0, 0)); for (int b = 0; b < burst; ++b) launch_fixed(vs[k].id); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); times[k].push_back(ms / burst);
        }
    printf("P=%lld margin=%.3f rounds=%d  (GB/s = 28 B/pt algorithmic)\n", (long long)P, margin, rounds);
    for (size_t k = 0; k < vs.size(); ++k) {
        std::sort(times[k].begin(), times[k].end());
        const float med = times[k][times[k].size() / 2], mn = times[k][0];
        printf("%-40s median %8.3f ms  min %8.3f ms  -> %7.1f GB/s (median)  %6.2f Gq/s\n", vs[k].name, med, mn, 28.0 * P / (med * 1e-3) / 1e9, P / (med * 1e-3) / 1e9);
    }
    return 0;
}
