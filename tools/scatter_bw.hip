// What does writing (val, grad) in CALLER order cost when the kernel walks the points in a sorted (Morton) order?
// (not part of the product; decides the write-back of the bucketed composed kernel)
//   A windows (configurations) x P points; out val[A][P] f32, grad[A][P][3] f32 = 16 B per (a, p) as in C4.
//   coalesced : thread i of window a writes slot i                     (what the unsorted kernel does)
//   scatter   : thread i of window a writes slot perm[i], perm random  (4-B + 12-B stores to unrelated lines)
//   scatter-xcd: same, with all blocks of one window on one XCD (block b runs on XCD b % 8)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/scatter_bw.hip -o tools/scatter_bw.bin && ./tools/scatter_bw.bin [A=200] [P=262144]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0 coalesced, 1 scatter, 2 scatter with window->XCD pinning, 3 scatter + nontemporal, 4 dword + dwordx3, 5 one 16-B packed record
__global__ __launch_bounds__(256) void write_k(const int* __restrict__ perm, int P, int A, float* __restrict__ val, float* __restrict__ grad) {
    int a, blk;
    const int bpw = P / 256;  // blocks per window
    if (MODE == 2) {
        const int b = blockIdx.x, xcd = b & 7, q = b >> 3;  // q-th block of this XCD
        a = (q / bpw) * 8 + xcd;
        blk = q % bpw;
        if (a >= A) return;
    } else {
        a = blockIdx.y;
        blk = blockIdx.x;
    }
    const int i = blk * 256 + threadIdx.x;
    const int j = MODE == 0 ? i : perm[i];
    const int64_t o = (int64_t)a * P + j;
    const float v = (float)i;
    if (MODE == 4) {
        val[o] = v;
        *reinterpret_cast<f32x3*>(grad + 3 * o) = f32x3{v + 1.f, v + 2.f, v + 3.f};
    } else if (MODE == 5) {
        reinterpret_cast<f32x4*>(grad)[o] = f32x4{v, v + 1.f, v + 2.f, v + 3.f};  // grad buffer reused as [A][P] x 16 B (needs 4/3 of its size: val follows it)
    } else if (MODE == 3) {
        __builtin_nontemporal_store(v, val + o);
        __builtin_nontemporal_store(v + 1.f, grad + 3 * o);
        __builtin_nontemporal_store(v + 2.f, grad + 3 * o + 1);
        __builtin_nontemporal_store(v + 3.f, grad + 3 * o + 2);
    } else {
        val[o] = v;
        grad[3 * o] = v + 1.f;
        grad[3 * o + 1] = v + 2.f;
        grad[3 * o + 2] = v + 3.f;
    }
}

int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 200;
    const int P = argc > 2 ? atoi(argv[2]) : 262144;
    std::vector<int> perm(P);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    int* dperm;
    float *val, *grad;
    CK(hipMalloc(&dperm, sizeof(int) * P));
    CK(hipMemcpy(dperm, perm.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    CK(hipMalloc(&val, sizeof(float) * (size_t)A * P));
    CK(hipMalloc(&grad, sizeof(float) * 3 * (size_t)A * P));
    float* grad4;
    CK(hipMalloc(&grad4, sizeof(float) * 4 * (size_t)A * P));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[6] = {"coalesced (4 dword stores/thread)", "scatter", "scatter, window pinned to an XCD", "scatter, nontemporal", "scatter, dword + dwordx3", "scatter, one packed 16-B record"};
    const double bytes = 16.0 * A * P;
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 6; ++mode) {
            std::vector<float> ms;
            for (int rep = 0; rep < 8; ++rep) {
                CK(hipEventRecord(e0));
                const int bpw = P / 256;
                if (mode == 0) hipLaunchKernelGGL(write_k<0>, dim3(bpw, A), dim3(256), 0, 0, dperm, P, A, val, grad);
                if (mode == 1) hipLaunchKernelGGL(write_k<1>, dim3(bpw, A), dim3(256), 0, 0, dperm, P, A, val, grad);
                if (mode == 2) hipLaunchKernelGGL(write_k<2>, dim3(bpw * ((A + 7) / 8) * 8), dim3(256), 0, 0, dperm, P, A, val, grad);
                if (mode == 4) hipLaunchKernelGGL(write_k<4>, dim3(bpw, A), dim3(256), 0, 0, dperm, P, A, val, grad);
                if (mode == 5) hipLaunchKernelGGL(write_k<5>, dim3(bpw, A), dim3(256), 0, 0, dperm, P, A, val, grad4);
                if (mode == 3) hipLaunchKernelGGL(write_k<3>, dim3(bpw, A), dim3(256), 0, 0, dperm, P, A, val, grad);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            if (round == 1) printf("%-40s median %.4f ms  %.0f GB/s\n", names[mode], ms[ms.size() / 2], bytes / ms[ms.size() / 2] / 1e6);
        }
    return 0;
}
