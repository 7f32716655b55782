// HBM ceiling of this box for the access mixes the query kernels have (not part of the product):
//   copy 1:1   float4 in -> float4 out          (MI355X_MICROARCH.md quotes 6.29 TB/s for this)
//   mix 12:16  12 B read + 16 B written per point (the CachedSDF query's algorithmic mix)
//   read-only / write-only
// each with plain and non-temporal accesses, at a capped grid (2048 x 256, grid-stride) and at one-16-B-per-thread.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/membw.hip -o tools/membw.bin && ./tools/membw.bin [MiB=1792]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void copy_k(const f32x4* __restrict__ in, f32x4* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// n_in float4 read, n_out float4 written (n_out >= n_in): the first n_in are copies, the rest constants
template <bool NT>
__global__ __launch_bounds__(256) void mix_k(const f32x4* __restrict__ in, int64_t n_in, f32x4* __restrict__ out, int64_t n_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t; i < n_in; i += stride) {
        const f32x4 a = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT) __builtin_nontemporal_store(a, out + i); else out[i] = a;
    }
    for (int64_t i = n_in + t; i < n_out; i += stride) {
        if (NT) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, out + i); else out[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void read_k(const f32x4* __restrict__ in, int64_t n, float* sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += NT ? __builtin_nontemporal_load(in + i) : in[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) *sink = 1.f;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_k(f32x4* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, out + i); else out[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    }
}

int main(int argc, char** argv) {
    const int64_t mib = argc > 1 ? atoll(argv[1]) : 1792;  // bytes moved per launch (read + written); 1792 MiB = the 64M-point query
    const int64_t total = mib << 20;
    f32x4 *a, *b;
    float* sink;
    CK(hipMalloc(&a, total));
    CK(hipMalloc(&b, total));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, total));
    CK(hipMemset(b, 2, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct V { const char* name; int id; };
    std::vector<V> vs = {{"copy 1:1 plain  grid 2048", 0}, {"copy 1:1 nt     grid 2048", 1}, {"copy 1:1 nt x4  grid 2048", 2},
                         {"copy 1:1 nt     grid 8192", 3}, {"copy 1:1 nt     1 float4/thread", 4}, {"copy 1:1 plain  1 float4/thread", 5},
                         {"mix 12:16 nt    grid 2048", 6}, {"mix 12:16 plain grid 2048", 7}, {"mix 12:16 nt    grid 8192", 8},
                         {"read-only nt    grid 2048", 9}, {"read-only plain grid 2048", 10}, {"write-only nt   grid 2048", 11},
                         {"write-only plain grid 2048", 12}, {"mix 12:16 nt    1 float4/thread", 13}, {"read-only nt    1 float4/thread", 14},
                         {"write-only nt   1 float4/thread", 15}, {"write-only plain 1 float4/thread", 16}, {"copy 1:1 nt     grid 65536", 17},
                         {"mix 12:16 nt    grid 65536", 18}, {"mix 12:16 nt    grid 16384", 19}};
    const int64_t half = total / 2 / 16;          // float4 count of each side for the 1:1 copy
    const int64_t n16 = total / 16;               // float4 count when everything is one side
    const int64_t pts = total / 28;               // points of the 12:16 mix
    const int64_t n_in = pts * 12 / 16, n_out = pts;
    printf("bytes per launch: %lld MiB (read + written)\n", (long long)mib);
    for (int round = 0; round < 2; ++round) {
        for (auto& v : vs) {
            std::vector<float> ms;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                switch (v.id) {
                    case 0: hipLaunchKernelGGL((copy_k<false, 1>), dim3(2048), dim3(256), 0, 0, a, b, half); break;
                    case 1: hipLaunchKernelGGL((copy_k<true, 1>), dim3(2048), dim3(256), 0, 0, a, b, half); break;
                    case 2: hipLaunchKernelGGL((copy_k<true, 4>), dim3(2048), dim3(256), 0, 0, a, b, half); break;
                    case 3: hipLaunchKernelGGL((copy_k<true, 1>), dim3(8192), dim3(256), 0, 0, a, b, half); break;
                    case 4: hipLaunchKernelGGL((copy_k<true, 1>), dim3((unsigned)((half + 255) / 256)), dim3(256), 0, 0, a, b, half); break;
                    case 5: hipLaunchKernelGGL((copy_k<false, 1>), dim3((unsigned)((half + 255) / 256)), dim3(256), 0, 0, a, b, half); break;
                    case 6: hipLaunchKernelGGL((mix_k<true>), dim3(2048), dim3(256), 0, 0, a, n_in, b, n_out); break;
                    case 7: hipLaunchKernelGGL((mix_k<false>), dim3(2048), dim3(256), 0, 0, a, n_in, b, n_out); break;
                    case 8: hipLaunchKernelGGL((mix_k<true>), dim3(8192), dim3(256), 0, 0, a, n_in, b, n_out); break;
                    case 9: hipLaunchKernelGGL((read_k<true>), dim3(2048), dim3(256), 0, 0, a, n16, sink); break;
                    case 10: hipLaunchKernelGGL((read_k<false>), dim3(2048), dim3(256), 0, 0, a, n16, sink); break;
                    case 11: hipLaunchKernelGGL((write_k<true>), dim3(2048), dim3(256), 0, 0, b, n16); break;
                    case 12: hipLaunchKernelGGL((write_k<false>), dim3(2048), dim3(256), 0, 0, b, n16); break;
                    case 13: hipLaunchKernelGGL((mix_k<true>), dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, 0, a, n_in, b, n_out); break;
                    case 14: hipLaunchKernelGGL((read_k<true>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, a, n16, sink); break;
                    case 15: hipLaunchKernelGGL((write_k<true>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, b, n16); break;
                    case 16: hipLaunchKernelGGL((write_k<false>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, b, n16); break;
                    case 17: hipLaunchKernelGGL((copy_k<true, 1>), dim3(65536), dim3(256), 0, 0, a, b, half); break;
                    case 18: hipLaunchKernelGGL((mix_k<true>), dim3(65536), dim3(256), 0, 0, a, n_in, b, n_out); break;
                    case 19: hipLaunchKernelGGL((mix_k<true>), dim3(16384), dim3(256), 0, 0, a, n_in, b, n_out); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            double bytes = (double)total;
            if ((v.id >= 6 && v.id <= 8) || v.id == 13 || v.id == 18 || v.id == 19) bytes = (double)(n_in + n_out) * 16;
            if (round == 1) printf("%-34s median %.4f ms  %.0f GB/s   (min %.4f ms  %.0f GB/s)\n", v.name, ms[ms.size() / 2], bytes / ms[ms.size() / 2] / 1e6, ms[0], bytes / ms[0] / 1e6);
        }
    }
    return 0;
}
