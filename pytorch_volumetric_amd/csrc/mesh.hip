// Brute-force point x triangle kernels (BASELINE configs C1, C5; also the voxel-cache build and LOOKUP_GT_SDF).
//   mesh_query:   closest surface point + ray-hit parity sign + gradient + face id   (reference sdf.py:122-172,
//                 where it is a device->host copy, two Embree BVH traversals on CPU threads, ~12 numpy passes)
//   chamfer_mesh: per-transform sum of (scale*d)^2 over the transformed points         (reference chamfer.py:79-94)
// fp32-VALU bound (~120 flop per point-triangle pair), not HBM bound: triangles are staged through LDS in tiles
// shared by the 4 waves of a block and read back as wave-uniform broadcasts; each lane owns PTS points so a tile
// read is amortised over PTS pairs.
#include "common.h"
#include "mesh_math.h"

namespace pvamd {

constexpr int kTile = 256;  // triangles per LDS tile: 256 * 36 B = 9 KB

struct MeshArgs {
    const float* tri;
    const float* normal;
    int F;
    double ray_dir[3];
};

PVAMD_DEV void stage_tile(float* __restrict__ lds, const float* __restrict__ tri, int f0, int F) {
    // 9 floats per triangle, contiguous in HBM: a coalesced copy of min(kTile, F-f0)*9 dwords
    const int n = min(kTile, F - f0) * 9;
    const float* src = tri + (int64_t)f0 * 9;
    for (int k = threadIdx.x; k < n; k += blockDim.x) lds[k] = src[k];
}

PVAMD_DEV void load_tri(const float* __restrict__ lds, int j, V3& a, V3& b, V3& c) {
    const float* t = lds + 9 * j;  // wave-uniform address: LDS broadcast
    a = v3(t[0], t[1], t[2]);
    b = v3(t[3], t[4], t[5]);
    c = v3(t[6], t[7], t[8]);
}

// np.linalg.norm of a float32 3-vector (sdf.py:141): products and sums rounded separately, left to right
PVAMD_DEV float norm3_unfused(V3 g) {
    return sqrt_rn(add_rn(add_rn(mul_rn(g.x, g.x), mul_rn(g.y, g.y)), mul_rn(g.z, g.z)));
}

template <int PTS>
__global__ __launch_bounds__(256) void mesh_query_kernel(MeshArgs m, const float* __restrict__ pts, int64_t P,
                                                          uint64_t seed, int64_t index_base,
                                                          float* __restrict__ out_closest,
                                                          float* __restrict__ out_dist, float* __restrict__ out_grad,
                                                          int* __restrict__ out_face, float* __restrict__ out_normal) {
    __shared__ float tile[kTile * 9];
    // lane-interleaved point ownership keeps the AoS loads/stores of one wave within a contiguous span
    const int64_t base = (int64_t)blockIdx.x * (blockDim.x * PTS) + threadIdx.x;
    V3 p[PTS], dir[PTS], best_q[PTS];
    float best_d2[PTS];
    int best_f[PTS], hits[PTS];
    bool live[PTS];
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int64_t i = base + (int64_t)k * blockDim.x;
        live[k] = i < P;
        const int64_t ii = live[k] ? i : 0;
        p[k] = v3(pts[3 * ii], pts[3 * ii + 1], pts[3 * ii + 2]);
        dir[k] = jitter_dir(m.ray_dir, seed, index_base + ii);
        best_d2[k] = INFINITY;
        best_f[k] = -1;
        best_q[k] = v3(NAN, NAN, NAN);
        hits[k] = 0;
    }
    for (int f0 = 0; f0 < m.F; f0 += kTile) {
        __syncthreads();
        stage_tile(tile, m.tri, f0, m.F);
        __syncthreads();
        const int n = min(kTile, m.F - f0);
        for (int j = 0; j < n; ++j) {
            V3 a, b, c;
            load_tri(tile, j, a, b, c);
#pragma unroll
            for (int k = 0; k < PTS; ++k) {
                const V3 q = closest_point_triangle(p[k], a, b, c);
                const V3 g = sub(q, p[k]);
                const float d2 = dot(g, g);
                if (d2 < best_d2[k]) {  // strict: lowest face id wins ties
                    best_d2[k] = d2;
                    best_f[k] = f0 + j;
                    best_q[k] = q;
                }
                hits[k] += ray_hits_triangle(p[k], dir[k], a, b, c);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int64_t i = base + (int64_t)k * blockDim.x;
        if (!live[k]) continue;
        V3 g = sub(best_q[k], p[k]);                 // sdf.py:139
        float d = norm3_unfused(g);                  // :141
        if (d > 0.f) {                               // :143-144
            g = v3(div_rn(g.x, d), div_rn(g.y, d), div_rn(g.z, d));
        }
        if (hits[k] & 1) d = -d;                     // :154-155 inside: negative distance
        else g = v3(-g.x, -g.y, -g.z);               // :157 outside: point away from the surface
        const int f = best_f[k];
        if (fabsf(d) < 1e-3f && f >= 0) {            // :162-164 on the surface: use the face normal
            g = v3(m.normal[3 * f], m.normal[3 * f + 1], m.normal[3 * f + 2]);
        }
        if (out_closest) {
            out_closest[3 * i] = best_q[k].x;
            out_closest[3 * i + 1] = best_q[k].y;
            out_closest[3 * i + 2] = best_q[k].z;
        }
        out_dist[i] = d;
        out_grad[3 * i] = g.x;
        out_grad[3 * i + 1] = g.y;
        out_grad[3 * i + 2] = g.z;
        if (out_face) out_face[i] = f;
        if (out_normal) {                            // :169-171
            out_normal[3 * i] = f >= 0 ? m.normal[3 * f] : NAN;
            out_normal[3 * i + 1] = f >= 0 ? m.normal[3 * f + 1] : NAN;
            out_normal[3 * i + 2] = f >= 0 ? m.normal[3 * f + 2] : NAN;
        }
    }
}

PVAMD_DEV double block_sum(double v, double* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0) {
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += scratch[w];
    }
    return total;  // valid on thread 0
}

// grid: x = point tiles, y = transform b
template <int PTS>
__global__ __launch_bounds__(256) void chamfer_mesh_kernel(MeshArgs m, const float* __restrict__ W,
                                                            const float* __restrict__ pts, int64_t N, float scale,
                                                            double* __restrict__ out_sum) {
    __shared__ float tile[kTile * 9];
    __shared__ double scratch[4];
    const float* M = W + 16 * (int64_t)blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * (blockDim.x * PTS) + threadIdx.x;
    V3 x[PTS], best_q[PTS];
    float best_d2[PTS];
    bool live[PTS];
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int64_t i = base + (int64_t)k * blockDim.x;
        live[k] = i < N;
        const int64_t ii = live[k] ? i : 0;
        const float px = pts[3 * ii], py = pts[3 * ii + 1], pz = pts[3 * ii + 2];
        // chamfer.py:81-82 transform_points, k-ordered fma chain
        x[k] = v3(add_rn(fmaf(M[2], pz, fmaf(M[1], py, mul_rn(M[0], px))), M[3]),
                  add_rn(fmaf(M[6], pz, fmaf(M[5], py, mul_rn(M[4], px))), M[7]),
                  add_rn(fmaf(M[10], pz, fmaf(M[9], py, mul_rn(M[8], px))), M[11]));
        best_d2[k] = INFINITY;
        best_q[k] = v3(NAN, NAN, NAN);
    }
    for (int f0 = 0; f0 < m.F; f0 += kTile) {
        __syncthreads();
        stage_tile(tile, m.tri, f0, m.F);
        __syncthreads();
        const int n = min(kTile, m.F - f0);
        for (int j = 0; j < n; ++j) {
            V3 a, b, c;
            load_tri(tile, j, a, b, c);
#pragma unroll
            for (int k = 0; k < PTS; ++k) {
                const V3 q = closest_point_triangle(x[k], a, b, c);
                const V3 g = sub(q, x[k]);
                const float d2 = dot(g, g);
                if (d2 < best_d2[k]) {
                    best_d2[k] = d2;
                    best_q[k] = q;
                }
            }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        if (!live[k]) continue;
        const float sd = mul_rn(scale, norm3_unfused(sub(best_q[k], x[k])));  // chamfer.py:92
        acc += (double)mul_rn(sd, sd);
    }
    __syncthreads();
    const double total = block_sum(acc, scratch);
    if (threadIdx.x == 0) atomicAdd(out_sum + blockIdx.y, total);
}

__global__ void zero_f64_kernel(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

static MeshArgs mesh_args(const pvamd_mesh_t& mesh) {
    MeshArgs m;
    m.tri = mesh.tri;
    m.normal = mesh.normal;
    m.F = mesh.F;
    for (int d = 0; d < 3; ++d) m.ray_dir[d] = mesh.ray_dir[d];
    return m;
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_mesh_query(const pvamd_mesh_t* mesh, const float* points, int64_t P, uint64_t jitter_seed,
                                int64_t index_base, float* out_closest, float* out_dist, float* out_grad,
                                int32_t* out_face, float* out_normal, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!mesh || !out_dist || !out_grad) return PVAMD_E_NULL;
    if (mesh->F < 0) return PVAMD_E_SHAPE;
    if (!points || (mesh->F > 0 && (!mesh->tri || !mesh->normal))) return PVAMD_E_NULL;
    const MeshArgs m = mesh_args(*mesh);
    // enough blocks to fill 256 CUs decides how many points a lane owns
    if (P >= (int64_t)256 * 256 * 8) {
        constexpr int PTS = 2;
        const unsigned blocks = (unsigned)((P + 256 * PTS - 1) / (256 * PTS));
        hipLaunchKernelGGL((mesh_query_kernel<PTS>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, m, points, P,
                           jitter_seed, index_base, out_closest, out_dist, out_grad, out_face, out_normal);
    } else {
        const unsigned blocks = (unsigned)((P + 63) / 64);
        hipLaunchKernelGGL((mesh_query_kernel<1>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, m, points, P,
                           jitter_seed, index_base, out_closest, out_dist, out_grad, out_face, out_normal);
    }
    return (int)hipGetLastError();
}

extern "C" int pvamd_chamfer_mesh(const pvamd_mesh_t* mesh, const float* W, int32_t B, const float* points, int64_t N,
                                  float scale, double* out_sum, void* stream) {
    if (!mesh || !out_sum) return PVAMD_E_NULL;
    if (B < 0 || B > 65535 * 1024 || N < 0 || mesh->F < 0) return PVAMD_E_SHAPE;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_f64_kernel, dim3((B + 255) / 256), dim3(256), 0, s, out_sum, B);
    if (N == 0) return (int)hipGetLastError();
    if (!W || !points || (mesh->F > 0 && !mesh->tri)) return PVAMD_E_NULL;
    const MeshArgs m = mesh_args(*mesh);
    // y-dimension of a HIP grid is limited to 65535: walk B in slabs
    for (int32_t b0 = 0; b0 < B; b0 += 65535) {
        const int32_t nb = (B - b0) < 65535 ? (B - b0) : 65535;
        if ((int64_t)nb * N >= (int64_t)256 * 256 * 8) {
            constexpr int PTS = 2;
            const unsigned gx = (unsigned)((N + 256 * PTS - 1) / (256 * PTS));
            hipLaunchKernelGGL((chamfer_mesh_kernel<PTS>), dim3(gx, nb), dim3(256), 0, s, m, W + 16 * (int64_t)b0,
                               points, N, scale, out_sum + b0);
        } else {
            const unsigned gx = (unsigned)((N + 63) / 64);
            hipLaunchKernelGGL((chamfer_mesh_kernel<1>), dim3(gx, nb), dim3(64), 0, s, m, W + 16 * (int64_t)b0, points,
                               N, scale, out_sum + b0);
        }
    }
    return (int)hipGetLastError();
}
