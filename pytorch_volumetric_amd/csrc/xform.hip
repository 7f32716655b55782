// RobotSDF.set_joint_configuration's batched 4x4 contraction (reference model_to_sdf.py:104-113):
//   obj_to_link[s*A+a] = offset_inv[s] @ rigid_inverse(link_world[s*A+a])
// The 4x4x4 products run on the matrix cores: v_mfma_f32_4x4x1_16b_f32 multiplies 16 independent 4x4 blocks per
// wave, one k-slice per issue, so four chained issues give D = A @ B for 16 (s,a) pairs at once.  f32 MFMA is an
// exact k-ordered fmaf chain, i.e. bit-identical to the scalar statement in oracle/pvamd_oracle.c (matmul4).
// S*A is at most a few thousand matrices: this kernel is about exactness and keeping the stack on device, not speed.
#include "common.h"
#include "exact_math.h"

namespace pvamd {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// lane l: block = l / 4 (which matrix of the 16), j = l % 4.
//   A operand: lane holds A[i = j][k]   (one value per k-slice)
//   B operand: lane holds B[k][j]
//   D result : vgpr r of lane holds D[i = r][j]
__global__ __launch_bounds__(64) void transform_stack_kernel(const float* __restrict__ offset_inv,
                                                              const float* __restrict__ link_world, int S, int A,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int j = lane & 3;
    const int64_t total = (int64_t)S * A;
    int64_t idx = (int64_t)blockIdx.x * 16 + (lane >> 2);
    const bool live = idx < total;
    if (!live) idx = total - 1;  // MFMA needs the whole wave; dead blocks recompute the last matrix and skip the store
    const int s = (int)(idx / A);
    const float* L = link_world + 16 * idx;
    const float* O = offset_inv + 16 * (int64_t)s;
    // B = rigid_inverse(L): rows 0..2 = [R^T | -R^T t], row 3 = [0 0 0 1].  This lane needs column j of B.
    float bcol[4];
    if (j < 3) {
        bcol[0] = L[4 * j + 0];  // (R^T)[0][j] = R[j][0]
        bcol[1] = L[4 * j + 1];
        bcol[2] = L[4 * j + 2];
        bcol[3] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // -(R^T t)_i = -(R[0][i] t0 + R[1][i] t1 + R[2][i] t2), k-ordered fma chain
            bcol[i] = -fmaf(L[8 + i], L[11], fmaf(L[4 + i], L[7], mul_rn(L[i], L[3])));
        }
        bcol[3] = 1.f;
    }
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = O[4 * j + k];  // A[i = j][k]
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bcol[k], acc, 0, 0, 0);
    }
    if (live) {
        float* D = out + 16 * idx;
        D[0 + j] = acc[0];
        D[4 + j] = acc[1];
        D[8 + j] = acc[2];
        D[12 + j] = acc[3];
    }
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_transform_stack(const float* offset_inv, const float* link_world, int32_t S, int32_t A, float* out,
                                     void* stream) {
    if (!offset_inv || !link_world || !out) return PVAMD_E_NULL;
    if (S < 1 || A < 1) return PVAMD_E_SHAPE;
    const int64_t total = (int64_t)S * A;
    hipLaunchKernelGGL(transform_stack_kernel, dim3((unsigned)((total + 15) / 16)), dim3(64), 0, (hipStream_t)stream,
                       offset_inv, link_world, S, A, out);
    return (int)hipGetLastError();
}
