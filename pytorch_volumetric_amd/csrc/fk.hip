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

// ---- round 4: the whole of RobotSDF.set_joint_configuration (model_to_sdf.py:94-113) in ONE launch ----
// Round 3 needed an H2D copy of q, two stock elementwise launches (sin, cos), chain_fk_kernel and transform_stack_kernel:
// 0.048 ms per call, more than the README-size query it feeds (0.013 ms).  Here one 64-lane block takes 64 configurations:
// phase 1, lane = configuration: sin / cos of the joint values (emitted, so that the CPU oracle can be fed the very same
// numbers), the frame walk of chain_fk_kernel, the leaf frames' 3x4 matrices left in LDS;
// phase 2, lanes regrouped as 16 blocks x 4: out[s*A+a] = offset_inv[s] @ rigid_inverse(world[s,a]) on the f32 MFMA
// (v_mfma_f32_4x4x1_16b_f32, the statement of transform_stack_kernel -- bit-identical to the oracle's fma chains).
typedef float f32x4m __attribute__((ext_vector_type(4)));

// Four waves per block: the frame walk is wave 0's (lane = configuration); the sines / cosines in front of it and the MFMA
// chains behind it are spread over all four (A = 200: 17.3 -> 9.8 us per launch, A = 20: 10.2 -> 7.7; they were 7 + 32 serial
// iterations of one wave at 64 configurations).
constexpr int kConfigureThreads = 256;
__global__ __launch_bounds__(kConfigureThreads) void configure_chain_kernel(const pvamd_joint_t* __restrict__ joints, int F,
                                                             const float* __restrict__ q, int A, int M,
                                                             const float* __restrict__ offset_inv, int S,
                                                             float* __restrict__ sincos, float* __restrict__ scratch,
                                                             float* __restrict__ link_world, float* __restrict__ out) {
    extern __shared__ float leafm[];  // [S][12][64]: the leaf frames of this block's 64 configurations, then the joint table
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = blockIdx.x * 64;
    const int a = a0 + lane;
    const bool live = a < A;
    const int ar = live ? a : A - 1;  // dead lanes redo the last configuration (same bits) and write nothing global
    // the joint table through LDS: one coalesced read instead of a chain of F dependent scalar-cache misses (the walk is
    // serial: at A = 20 the kernel IS its latency)
    pvamd_joint_t* sj = reinterpret_cast<pvamd_joint_t*>(leafm + (size_t)S * 12 * 64);
    float* ssc = reinterpret_cast<float*>(sj + F);  // [64][M][2]: sin, cos of this block's joint values
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(joints);
        uint32_t* dst = reinterpret_cast<uint32_t*>(sj);
        const int words = F * (int)(sizeof(pvamd_joint_t) / 4);
        for (int w = threadIdx.x; w < words; w += kConfigureThreads) dst[w] = src[w];
    }
    // phase 0: every sine / cosine of the block, lanes = (configuration, joint) pairs -- ~100 instructions each that would
    // otherwise sit in every lane's serial frame walk (with A = 20 only 20 lanes walk; all 64 work here)
    {
        const int nA0 = A - a0 < 64 ? A - a0 : 64;
        for (int idx = threadIdx.x; idx < nA0 * M; idx += kConfigureThreads) {
            const float qv = q[(int64_t)a0 * M + idx];
            const float sv = sinf(qv), cv = cosf(qv);  // (joint values are a few radians: the small-argument path of both)
            ssc[2 * idx] = sv;
            ssc[2 * idx + 1] = cv;
            if (sincos) {
                sincos[((int64_t)a0 * M + idx) * 2] = sv;
                sincos[((int64_t)a0 * M + idx) * 2 + 1] = cv;
            }
        }
    }
    __syncthreads();
    float m[12];
    for (int f = 0; f < (wave == 0 ? F : 0); ++f) {
        const pvamd_joint_t& J = sj[f];  // wave-uniform LDS reads (broadcast)
        float P[12];
        if (J.parent < 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = (k % 5 == 0) ? 1.f : 0.f;
        } else if (J.parent == f - 1) {  // a serial chain: the parent is what this lane has just computed
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = m[k];
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = scratch[((int64_t)J.parent * 12 + k) * A + ar];  // own column: same thread wrote it
        }
        compose_affine(P, J.origin, m);
        if (J.jtype == 1) {  // revolute / continuous: Rodrigues rotation about the joint axis
            const float s = ssc[2 * ((ar - a0) * M + J.jcol)], c = ssc[2 * ((ar - a0) * M + J.jcol) + 1];
            const float x = J.axis[0], y = J.axis[1], z = J.axis[2];
            const float t = sub_rn(1.f, c);
            const float tx = mul_rn(t, x), ty = mul_rn(t, y), tz = mul_rn(t, z);
            float R[12];
            R[0] = fmaf(tx, x, c);              R[1] = fmaf(tx, y, -mul_rn(s, z)); R[2] = fmaf(tx, z, mul_rn(s, y));  R[3] = 0.f;
            R[4] = fmaf(tx, y, mul_rn(s, z));   R[5] = fmaf(ty, y, c);             R[6] = fmaf(ty, z, -mul_rn(s, x)); R[7] = 0.f;
            R[8] = fmaf(tx, z, -mul_rn(s, y));  R[9] = fmaf(ty, z, mul_rn(s, x));  R[10] = fmaf(tz, z, c);            R[11] = 0.f;
            float o[12];
            compose_affine(m, R, o);
#pragma unroll
            for (int k = 0; k < 12; ++k) m[k] = o[k];
        } else if (J.jtype == 2) {  // prismatic: translate along the joint axis
            const float d = q[(int64_t)ar * M + J.jcol];
            float T[12] = {1.f, 0.f, 0.f, mul_rn(J.axis[0], d), 0.f, 1.f, 0.f, mul_rn(J.axis[1], d),
                           0.f, 0.f, 1.f, mul_rn(J.axis[2], d)};
            float o[12];
            compose_affine(m, T, o);
#pragma unroll
            for (int k = 0; k < 12; ++k) m[k] = o[k];
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < 12; ++k) scratch[((int64_t)f * 12 + k) * A + a] = m[k];
        }
        if (J.leaf_slot >= 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) leafm[(J.leaf_slot * 12 + k) * 64 + lane] = m[k];
            if (link_world && live) {
                float* o = link_world + 16 * ((int64_t)J.leaf_slot * A + a);
#pragma unroll
                for (int k = 0; k < 12; ++k) o[k] = m[k];
                o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
            }
        }
    }
    __syncthreads();  // the leaf frames wave 0 left in LDS, for all four waves
    // phase 2: 16 (leaf, configuration) pairs per MFMA chain; lane l: pair = l / 4, j = l % 4
    const int nA = A - a0 < 64 ? A - a0 : 64;
    const int pairs = S * nA;
    const int j = lane & 3;
    for (int p0 = 16 * wave; p0 < pairs; p0 += 16 * (kConfigureThreads / 64)) {
        int p = p0 + (lane >> 2);
        const bool pl = p < pairs;
        if (!pl) p = pairs - 1;  // MFMA needs the whole wave
        const int s = p / nA, c = p % nA;
        const float* L = leafm + (s * 12) * 64 + c;  // L[k * 64] = element k of the 3x4 world matrix
        const float* O = offset_inv + 16 * (int64_t)s;
        float bcol[4];
        if (j < 3) {  // column j of rigid_inverse(world): (R^T)[i][j] = R[j][i]
            bcol[0] = L[(4 * j + 0) * 64];
            bcol[1] = L[(4 * j + 1) * 64];
            bcol[2] = L[(4 * j + 2) * 64];
            bcol[3] = 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)  // -(R^T t)_i, k-ordered fma chain (transform_stack_kernel's statement)
                bcol[i] = -fmaf(L[(8 + i) * 64], L[11 * 64], fmaf(L[(4 + i) * 64], L[7 * 64], mul_rn(L[i * 64], L[3 * 64])));
            bcol[3] = 1.f;
        }
        f32x4m acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(O[4 * j + k], bcol[k], acc, 0, 0, 0);
        if (pl) {
            float* D = out + 16 * ((int64_t)s * A + a0 + c);
            D[0 + j] = acc[0];
            D[4 + j] = acc[1];
            D[8 + j] = acc[2];
            D[12 + j] = acc[3];
        }
    }
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_configure_chain(const pvamd_joint_t* joints, int32_t F, const float* q, int32_t A, int32_t M,
                                     const float* offset_inv, int32_t S, float* sincos_out, float* scratch,
                                     float* link_world_out, float* stack_out, void* stream) {
    if (F < 1 || A < 1 || M < 0 || S < 1) return PVAMD_E_SHAPE;
    if (!joints || !offset_inv || !scratch || !stack_out) return PVAMD_E_NULL;
    if (M > 0 && !q) return PVAMD_E_NULL;
    const size_t lds = (size_t)S * 12 * 64 * sizeof(float) + (size_t)F * sizeof(pvamd_joint_t) + (size_t)64 * M * 2 * sizeof(float);
    if (lds > 150 * 1024) return PVAMD_E_SHAPE;  // ~50 SDF-carrying links: use pvamd_chain_fk + pvamd_transform_stack
    if (lds > 64 * 1024) {  // more dynamic LDS than a launch gets by default (21+ SDF-carrying links): opt in (hipFuncSetAttribute leaves a per-function attribute behind: harmless, a later call only ever raises it)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(configure_chain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(configure_chain_kernel, dim3((A + 63) / 64), dim3(kConfigureThreads), lds, (hipStream_t)stream, joints, F, q, A, M,
                       offset_inv, S, sincos_out, scratch, link_world_out, stack_out);
    return (int)hipGetLastError();
}

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
