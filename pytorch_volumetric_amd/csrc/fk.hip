// Forward kinematics of a URDF kinematic tree for A configurations at once, on device
// (reference: RobotSDF.set_joint_configuration, model_to_sdf.py:94-102, where pytorch_kinematics runs a chain of small
// host-driven torch ops per frame).  One lane per configuration walks the frames in topological order:
//     world[f] = world[parent[f]] @ origin[f] @ motion(joint f, q)
// and drops the matrices of the links that carry an SDF straight into the leaf-major stack that
// pvamd_transform_stack / pvamd_composed_query consume.  sin(q) / cos(q) come in as arrays (two stock elementwise
// launches upstream) so that everything here is exact fma chains and the CPU oracle can state the same sequence.
#include "common.h"
#include "exact_math.h"

namespace pvamd {

// affine 3x4 composition C = A @ B (implicit last row 0 0 0 1), k-ordered fma chains (same statement as matmul4 in
// the oracle restricted to the affine part)
PVAMD_DEV void compose_affine(const float* A, const float* B, float* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = mul_rn(A[4 * i], B[j]);
            acc = fmaf(A[4 * i + 1], B[4 + j], acc);
            acc = fmaf(A[4 * i + 2], B[8 + j], acc);
            if (j == 3) acc = add_rn(acc, A[4 * i + 3]);  // + A[i][3] * 1
            C[4 * i + j] = acc;
        }
    }
}

__global__ __launch_bounds__(64) void chain_fk_kernel(const pvamd_joint_t* __restrict__ joints, int F,
                                                      const float* __restrict__ q, const float* __restrict__ sin_q,
                                                      const float* __restrict__ cos_q, int A, int M,
                                                      float* __restrict__ scratch, float* __restrict__ link_world) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    for (int f = 0; f < F; ++f) {
        const pvamd_joint_t& J = joints[f];  // wave-uniform: scalar loads
        float P[12], m[12];
        if (J.parent < 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = (k % 5 == 0) ? 1.f : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = scratch[((int64_t)J.parent * 12 + k) * A + a];  // [F][12][A]: coalesced
        }
        compose_affine(P, J.origin, m);
        if (J.jtype == 1) {  // revolute / continuous: Rodrigues rotation about the joint axis
            const float s = sin_q[(int64_t)a * M + J.jcol], c = cos_q[(int64_t)a * M + J.jcol];
            const float x = J.axis[0], y = J.axis[1], z = J.axis[2];
            const float t = sub_rn(1.f, c);
            const float tx = mul_rn(t, x), ty = mul_rn(t, y), tz = mul_rn(t, z);
            float R[12];
            R[0] = fmaf(tx, x, c);              R[1] = fmaf(tx, y, -mul_rn(s, z)); R[2] = fmaf(tx, z, mul_rn(s, y));  R[3] = 0.f;
            R[4] = fmaf(tx, y, mul_rn(s, z));   R[5] = fmaf(ty, y, c);             R[6] = fmaf(ty, z, -mul_rn(s, x)); R[7] = 0.f;
            R[8] = fmaf(tx, z, -mul_rn(s, y));  R[9] = fmaf(ty, z, mul_rn(s, x));  R[10] = fmaf(tz, z, c);            R[11] = 0.f;
            float out[12];
            compose_affine(m, R, out);
#pragma unroll
            for (int k = 0; k < 12; ++k) m[k] = out[k];
        } else if (J.jtype == 2) {  // prismatic: translate along the joint axis
            const float d = q[(int64_t)a * M + J.jcol];
            float T[12] = {1.f, 0.f, 0.f, mul_rn(J.axis[0], d), 0.f, 1.f, 0.f, mul_rn(J.axis[1], d),
                           0.f, 0.f, 1.f, mul_rn(J.axis[2], d)};
            float out[12];
            compose_affine(m, T, out);
#pragma unroll
            for (int k = 0; k < 12; ++k) m[k] = out[k];
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) scratch[((int64_t)f * 12 + k) * A + a] = m[k];
        if (J.leaf_slot >= 0) {
            float* o = link_world + 16 * ((int64_t)J.leaf_slot * A + a);
#pragma unroll
            for (int k = 0; k < 12; ++k) o[k] = m[k];
            o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
        }
    }
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_chain_fk(const pvamd_joint_t* joints, int32_t F, const float* q, const float* sin_q,
                              const float* cos_q, int32_t A, int32_t M, float* scratch, float* link_world_out,
                              void* stream) {
    if (F < 1 || A < 1 || M < 0) return PVAMD_E_SHAPE;
    if (!joints || !scratch || !link_world_out) return PVAMD_E_NULL;
    if (M > 0 && (!q || !sin_q || !cos_q)) return PVAMD_E_NULL;
    hipLaunchKernelGGL(chain_fk_kernel, dim3((A + 63) / 64), dim3(64), 0, (hipStream_t)stream, joints, F, q, sin_q, cos_q,
                       A, M, scratch, link_world_out);
    return (int)hipGetLastError();
}
