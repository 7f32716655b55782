// Point x triangle kernels (BASELINE configs C1, C5; also the voxel-cache build and LOOKUP_GT_SDF).
//   mesh_query:   closest surface point + ray-hit parity sign + gradient + face id   (reference sdf.py:122-172,
//                 where it is a device->host copy, two Embree BVH traversals on CPU threads, ~12 numpy passes)
//   chamfer_mesh: per-transform sum of (scale*d)^2 over the transformed points         (reference chamfer.py:79-94)
//
// fp32-VALU bound, not HBM bound.  Structure:
//   * pvamd_mesh_prepare turns the soup into 112-byte records (corners, edge vectors, geometric normal, bounding
//     sphere, original face id) + one bounding sphere per tile of 256 records.
//   * a block owns 64 points (one per lane) and SLICES waves; a tile of records is staged into LDS once per block and
//     its triangles are dealt round-robin to the waves (small P -> many slices so the 1024 SIMDs still fill).
//   * two-level conservative culling: a tile is skipped when, for every lane of the block, its sphere is farther than
//     the lane's current best distance AND misses the lane's ray; inside a live tile the same test per triangle,
//     wave-uniform.  A skipped triangle provably cannot lower a lane's best d^2 nor be hit by its ray, and the exact
//     tests run on every lane whenever any lane needs them, so results are bit-identical to the plain double loop of
//     oracle/pvamd_oracle.c.  With spatially sorted triangles and points this is a flat two-level BVH.
//   * ties in d^2 resolve to the lowest ORIGINAL face id (lexicographic min), independent of processing order.
#include "common.h"
#include "mesh_math.h"

namespace pvamd {

constexpr int kRec = PVAMD_TRI_REC;    // floats per record
constexpr int kTile = PVAMD_TRI_TILE;  // records per tile: 256 * 112 B = 28 KB of LDS
// record layout (float index): 0-2 ctr, 3 r | 4-6 a, 7 face id bits | 8-10 b | 12-14 c | 16-18 ab | 20-22 ac | 24-26 Ng

struct MeshArgs {
    const float* normal;
    const float* rec;
    const float* tiles;
    const int* rec_of_face;
    int F;
    double ray_dir[3];
};

// ---------------------------------------------------------------------------------------------------------------
// prepare
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mesh_prepare_records(const float* __restrict__ tri, const int* __restrict__ face_id,
                                                            int F, float abs_margin, float* __restrict__ rec) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float* t = tri + 9 * (int64_t)f;
    const V3 a = v3(t[0], t[1], t[2]), b = v3(t[3], t[4], t[5]), c = v3(t[6], t[7], t[8]);
    const V3 ab = sub(b, a), ac = sub(c, a);
    const V3 Ng = cross(ac, sub(a, b));  // Embree: e1 = v0 - v1, e2 = v2 - v0, Ng = cross(e2, e1)
    // bounding sphere: centroid + largest corner distance, inflated so that round-off can never make it too small
    const V3 ctr = v3((a.x + b.x + c.x) * (1.f / 3.f), (a.y + b.y + c.y) * (1.f / 3.f), (a.z + b.z + c.z) * (1.f / 3.f));
    const V3 da = sub(a, ctr), db = sub(b, ctr), dc = sub(c, ctr);
    const float r2 = fmaxf(dot(da, da), fmaxf(dot(db, db), dot(dc, dc)));
    const float r = sqrt_rn(r2) * 1.00001f + abs_margin;
    float* o = rec + (int64_t)kRec * f;
    o[0] = ctr.x; o[1] = ctr.y; o[2] = ctr.z; o[3] = r;
    o[4] = a.x; o[5] = a.y; o[6] = a.z; o[7] = __int_as_float(face_id ? face_id[f] : f);
    o[8] = b.x; o[9] = b.y; o[10] = b.z; o[11] = 0.f;
    o[12] = c.x; o[13] = c.y; o[14] = c.z; o[15] = 0.f;
    o[16] = ab.x; o[17] = ab.y; o[18] = ab.z; o[19] = 0.f;
    o[20] = ac.x; o[21] = ac.y; o[22] = ac.z; o[23] = 0.f;
    o[24] = Ng.x; o[25] = Ng.y; o[26] = Ng.z; o[27] = 0.f;
}

// one block per tile: sphere around the mean of the member centres, radius = max(|c_i - mean| + r_i), inflated
__global__ __launch_bounds__(256) void mesh_prepare_tiles(const float* __restrict__ rec, int F, float abs_margin,
                                                          float* __restrict__ tiles) {
    __shared__ float sh[4][4];
    const int tile = blockIdx.x;
    const int f = tile * kTile + threadIdx.x;
    const bool live = f < F;
    const float* o = rec + (int64_t)kRec * (live ? f : (F - 1));
    float cx = live ? o[0] : 0.f, cy = live ? o[1] : 0.f, cz = live ? o[2] : 0.f, n = live ? 1.f : 0.f;
    float sx = cx, sy = cy, sz = cz, sn = n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_down(sx, off, 64); sy += __shfl_down(sy, off, 64);
        sz += __shfl_down(sz, off, 64); sn += __shfl_down(sn, off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[wave][0] = sx; sh[wave][1] = sy; sh[wave][2] = sz; sh[wave][3] = sn; }
    __syncthreads();
    const float tn = sh[0][3] + sh[1][3] + sh[2][3] + sh[3][3];
    const float mx = (sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0]) / tn;
    const float my = (sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1]) / tn;
    const float mz = (sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2]) / tn;
    __syncthreads();
    float reach = 0.f;
    if (live) {
        const float dx = cx - mx, dy = cy - my, dz = cz - mz;
        reach = sqrt_rn(dx * dx + dy * dy + dz * dz) * 1.00001f + o[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) reach = fmaxf(reach, __shfl_down(reach, off, 64));
    if (lane == 0) sh[wave][0] = reach;
    __syncthreads();
    if (threadIdx.x == 0) {
        float* t = tiles + 4 * (int64_t)tile;
        t[0] = mx; t[1] = my; t[2] = mz;
        t[3] = fmaxf(fmaxf(sh[0][0], sh[1][0]), fmaxf(sh[2][0], sh[3][0])) * 1.00001f + abs_margin;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// query
// ---------------------------------------------------------------------------------------------------------------
// np.linalg.norm of a float32 3-vector (sdf.py:141): products and sums rounded separately, left to right
PVAMD_DEV float norm3_unfused(V3 g) {
    return sqrt_rn(add_rn(add_rn(mul_rn(g.x, g.x), mul_rn(g.y, g.y)), mul_rn(g.z, g.z)));
}

// Ericson's closest point with the triangle's edge vectors precomputed (same operations as closest_point_triangle)
PVAMD_DEV V3 closest_point_prepared(V3 p, V3 a, V3 b, V3 c, V3 ab, V3 ac) {
    const V3 ap = sub(p, a);
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) return a;
    const V3 bp = sub(p, b);
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) return b;
    const V3 cp = sub(p, c);
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) return c;
    const float vc = fmaf(d1, d4, -mul_rn(d3, d2));
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) return madd(div_rn(d1, sub_rn(d1, d3)), ab, a);
    const float vb = fmaf(d5, d2, -mul_rn(d1, d6));
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) return madd(div_rn(d2, sub_rn(d2, d6)), ac, a);
    const float va = fmaf(d3, d6, -mul_rn(d5, d4));
    const float d43 = sub_rn(d4, d3), d56 = sub_rn(d5, d6);
    if (va <= 0.f && d43 >= 0.f && d56 >= 0.f) return madd(div_rn(d43, add_rn(d43, d56)), sub(c, b), b);
    const float denom = div_rn(1.f, add_rn(add_rn(va, vb), vc));
    return madd(mul_rn(vc, denom), ac, madd(mul_rn(vb, denom), ab, a));
}

// Embree's Moeller-Trumbore test with the geometric normal precomputed (same operations as ray_hits_triangle:
// e1 = a - b = -ab and C = a - org = -ap are exact negations, so the dot products below are the same numbers)
PVAMD_DEV int ray_hits_prepared(V3 org, V3 dir, V3 a, V3 ab, V3 ac, V3 Ng) {
    const V3 e1 = v3(-ab.x, -ab.y, -ab.z);
    const V3 C = sub(a, org);
    const V3 R = cross(C, dir);
    const float den = dot(Ng, dir);
    const float absden = fabsf(den);
    const float sgn = den < 0.f ? -1.f : 1.f;
    const float U = mul_rn(dot(R, ac), sgn);
    const float V = mul_rn(dot(R, e1), sgn);
    const float T = mul_rn(dot(Ng, C), sgn);
    return ((den != 0.f) && (U >= 0.f) && (V >= 0.f) && (add_rn(U, V) <= absden) && (T > 0.f)) ? 1 : 0;
}

struct LaneState {
    V3 p;
    float best_d2, reach, reach2;  // reach = inflated sqrt(best_d2): "closer than this could still win"
    int best_f;
};

PVAMD_DEV void set_best(LaneState& s, float d2, int f) {
    s.best_d2 = d2;
    s.best_f = f;
    const float reach = sqrt_rn(d2) * 1.00001f;  // NaN/inf propagate: comparisons against them keep the triangle
    if (!(reach >= s.reach)) {                   // never loosen a bound that is already tighter (seeded, below)
        s.reach = reach;
        s.reach2 = reach * reach;
    }
}

// Upper bound on the distance from p to the mesh before any triangle is looked at: every tile sphere contains at
// least one whole triangle, so min over tiles of (|p - ctr| + r) bounds the nearest-triangle distance from above.
// Seeding `reach` with it lets the very first tiles be culled (they are visited in storage order, not nearest-first).
// Also returns the tile that attains the bound for this lane: visiting the nearest tile first tightens `reach` to the
// true distance immediately, after which almost every other tile fails the sphere test.
PVAMD_DEV int seed_reach(const MeshArgs& m, LaneState& s) {
    const int ntiles = (m.F + kTile - 1) / kTile;
    float bound = INFINITY;
    int nearest = 0;
    for (int ti = 0; ti < ntiles; ++ti) {
        const float* ts = m.tiles + 4 * (int64_t)ti;
        const V3 w = v3(ts[0] - s.p.x, ts[1] - s.p.y, ts[2] - s.p.z);
        const float b = sqrt_rn(dot(w, w)) * 1.00001f + ts[3];
        if (b < bound) {  // a NaN point never passes: keeps INFINITY / tile 0
            bound = b;
            nearest = ti;
        }
    }
    s.reach = bound * 1.00001f;
    s.reach2 = s.reach * s.reach;
    return nearest;
}

// sphere (ctr, r) cannot contain a point closer to p than the current best:  |p-ctr| > r + reach
PVAMD_DEV bool sphere_may_improve(const LaneState& s, float dist2, float r) {
    const float bound = fmaf(2.f * s.reach, r, fmaf(r, r, s.reach2));  // (r + reach)^2
    return !(dist2 > bound * 1.00001f);
}

// sphere (ctr, r) may be crossed by the ray p + t*dn, t > 0 (dn unit):  distance from ctr to the line <= r, not behind p
PVAMD_DEV bool sphere_may_hit(float dist2, float tp, float r) {
    const float perp2 = fmaf(-tp, tp, dist2);
    return !(perp2 > fmaf(r, r, 2e-6f * dist2)) && !(tp < -r);
}

template <int PG, int SLICES, bool WITH_RAY>
PVAMD_DEV void scan_mesh(const MeshArgs& m, float* __restrict__ tile_lds, LaneState& s, V3 dir, V3 dn, int& hits,
                         int* __restrict__ ctl_lds) {
    constexpr int kWaves = PG * SLICES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slice = wave % SLICES;
    const int ntiles = (m.F + kTile - 1) / kTile;
    // Visit order: the tiles nearest to the first and the last point of the block (the two ends of a Morton-sorted run
    // of points), then storage order -- the same for every wave of the block.
    const int my_nearest = seed_reach(m, s);
    int first0 = my_nearest, first1 = my_nearest;
    if (kWaves == 1) {
        first0 = __shfl(my_nearest, 0, 64);
        first1 = __shfl(my_nearest, 63, 64);
    } else {
        if (wave == 0 && lane == 0) ctl_lds[1] = my_nearest;
        if (wave == kWaves - 1 && lane == 63) ctl_lds[2] = my_nearest;
        __syncthreads();
        first0 = ctl_lds[1];
        first1 = ctl_lds[2];
    }
    for (int step = 0; step < ntiles + 2; ++step) {
        int ti;
        if (step == 0) ti = first0;
        else if (step == 1) { if (first1 == first0) continue; ti = first1; }
        else { ti = step - 2; if (ti == first0 || ti == first1) continue; }
        // ---- tile-level cull (block-uniform decision: every wave must agree before the barrier) ----
        const float* ts = m.tiles + 4 * (int64_t)ti;  // uniform address: scalar loads
        const V3 wt = v3(ts[0] - s.p.x, ts[1] - s.p.y, ts[2] - s.p.z);
        const float tdist2 = dot(wt, wt);
        bool need = sphere_may_improve(s, tdist2, ts[3]);
        if (WITH_RAY) need = need || sphere_may_hit(tdist2, dot(wt, dn), ts[3]);
        if (kWaves == 1) {
            if (!__any(need)) continue;
            __syncthreads();
        } else {
            __syncthreads();  // previous tile fully consumed; flag reusable
            if (threadIdx.x == 0) ctl_lds[0] = 0;
            __syncthreads();
            if (__any(need) && lane == 0) atomicOr(ctl_lds, 1);
            __syncthreads();
            if (ctl_lds[0] == 0) continue;
        }
        // ---- stage the tile: contiguous float4 copy ----
        const int n = min(kTile, m.F - ti * kTile);
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(m.rec + (int64_t)kRec * kTile * ti);
            f32x4_alias* dst = reinterpret_cast<f32x4_alias*>(tile_lds);
            for (int k = threadIdx.x; k < n * (kRec / 4); k += blockDim.x) dst[k] = src[k];
        }
        __syncthreads();
        if (kWaves > 1 && !__any(need)) continue;  // this wave's 64 points do not need the tile another wave asked for
        // ---- triangles of this tile, dealt round-robin to the slices of a point group ----
        for (int j = slice; j < n; j += SLICES) {
            const float* o = tile_lds + kRec * j;  // wave-uniform address: LDS broadcast reads
            const V3 w = v3(o[0] - s.p.x, o[1] - s.p.y, o[2] - s.p.z);
            const float r = o[3];
            const float dist2 = dot(w, w);
            const bool need_c = sphere_may_improve(s, dist2, r);
            bool need_r = false;
            if (WITH_RAY) need_r = sphere_may_hit(dist2, dot(w, dn), r);
            if (!__any(need_c || need_r)) continue;
            const V3 a = v3(o[4], o[5], o[6]);
            const V3 ab = v3(o[16], o[17], o[18]), ac = v3(o[20], o[21], o[22]);
            if (__any(need_c)) {
                const V3 b = v3(o[8], o[9], o[10]), c = v3(o[12], o[13], o[14]);
                const int f = __float_as_int(o[7]);
                const V3 q = closest_point_prepared(s.p, a, b, c, ab, ac);
                const V3 g = sub(q, s.p);
                const float d2 = dot(g, g);
                if (d2 < s.best_d2 || (d2 == s.best_d2 && f < s.best_f)) set_best(s, d2, f);
            }
            if (WITH_RAY) {
                if (__any(need_r)) hits += ray_hits_prepared(s.p, dir, a, ab, ac, v3(o[24], o[25], o[26]));
            }
        }
    }
}

// merge the slices of each point group: (d2, face) by lexicographic min, hit counts by sum; valid in slice 0
template <int PG, int SLICES>
PVAMD_DEV void merge_slices(float* __restrict__ scratch, LaneState& s, int& hits) {
    if (SLICES == 1) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pg = wave / SLICES, slice = wave % SLICES;
    __syncthreads();  // tile LDS no longer needed: reuse it as scratch [waves][3][64]
    scratch[(wave * 3 + 0) * 64 + lane] = s.best_d2;
    scratch[(wave * 3 + 1) * 64 + lane] = __int_as_float(s.best_f);
    scratch[(wave * 3 + 2) * 64 + lane] = __int_as_float(hits);
    __syncthreads();
    if (slice == 0) {
        for (int k = 1; k < SLICES; ++k) {
            const int w = pg * SLICES + k;
            const float d2 = scratch[(w * 3 + 0) * 64 + lane];
            const int f = __float_as_int(scratch[(w * 3 + 1) * 64 + lane]);
            hits += __float_as_int(scratch[(w * 3 + 2) * 64 + lane]);
            if (f >= 0 && (s.best_f < 0 || d2 < s.best_d2 || (d2 == s.best_d2 && f < s.best_f))) {
                s.best_d2 = d2;
                s.best_f = f;
            }
        }
    }
}

// the closest point on the winning face, recomputed from its corners (same operations as during the scan)
PVAMD_DEV V3 closest_on_face(const MeshArgs& m, const float* __restrict__ tri_of_face, V3 p) {
    const float* o = tri_of_face;
    return closest_point_prepared(p, v3(o[4], o[5], o[6]), v3(o[8], o[9], o[10]), v3(o[12], o[13], o[14]),
                                  v3(o[16], o[17], o[18]), v3(o[20], o[21], o[22]));
}

template <int PG, int SLICES>
__global__ __launch_bounds__(64 * PG * SLICES) void mesh_query_kernel(MeshArgs m, const int* __restrict__ order,
                                                                const float* __restrict__ pts, int64_t P,
                                                                uint64_t seed, int64_t index_base,
                                                                float* __restrict__ out_closest,
                                                                float* __restrict__ out_dist,
                                                                float* __restrict__ out_grad,
                                                                int* __restrict__ out_face,
                                                                float* __restrict__ out_normal) {
    __shared__ __attribute__((aligned(16))) float tile_lds[kTile * kRec];
    __shared__ int ctl[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = ((int64_t)blockIdx.x * PG + wave / SLICES) * 64 + lane;  // position in processing order
    const bool live = k < P;
    const int64_t kk = live ? k : (P - 1);
    const int64_t i = order ? (int64_t)order[kk] : kk;  // the point this lane owns (spatially sorted processing)
    const int64_t ii = i;
    LaneState s;
    s.p = v3(pts[3 * ii], pts[3 * ii + 1], pts[3 * ii + 2]);
    s.best_f = -1;
    s.best_d2 = INFINITY;
    s.reach = INFINITY;
    s.reach2 = INFINITY;
    const V3 dir = jitter_dir(m.ray_dir, seed, index_base + ii);
    const float inv_len = 1.f / sqrt_rn(dot(dir, dir));
    const V3 dn = v3(dir.x * inv_len, dir.y * inv_len, dir.z * inv_len);
    int hits = 0;
    scan_mesh<PG, SLICES, true>(m, tile_lds, s, dir, dn, hits, ctl);
    merge_slices<PG, SLICES>(tile_lds, s, hits);
    if ((wave % SLICES) != 0 || !live) return;

    const int f = s.best_f;
    V3 q = v3(NAN, NAN, NAN);
    if (f >= 0) q = closest_on_face(m, m.rec + (int64_t)kRec * m.rec_of_face[f], s.p);
    V3 g = sub(q, s.p);                          // sdf.py:139
    float d = norm3_unfused(g);                  // :141
    if (d > 0.f) g = v3(div_rn(g.x, d), div_rn(g.y, d), div_rn(g.z, d));  // :143-144
    if (hits & 1) d = -d;                        // :154-155 inside: negative distance
    else g = v3(-g.x, -g.y, -g.z);               // :157 outside: point away from the surface
    if (fabsf(d) < 1e-3f && f >= 0) {            // :162-164 on the surface: use the face normal
        g = v3(m.normal[3 * f], m.normal[3 * f + 1], m.normal[3 * f + 2]);
    }
    if (out_closest) {
        out_closest[3 * i] = q.x;
        out_closest[3 * i + 1] = q.y;
        out_closest[3 * i + 2] = q.z;
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

// grid: x = tiles of 64 points, y = transform b
template <int PG, int SLICES>
__global__ __launch_bounds__(64 * PG * SLICES) void chamfer_mesh_kernel(MeshArgs m, const int* __restrict__ order,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ pts, int64_t N, float scale,
                                                                  double* __restrict__ out_sum) {
    __shared__ __attribute__((aligned(16))) float tile_lds[kTile * kRec];
    __shared__ int ctl[4];
    const float* M = W + 16 * (int64_t)blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = ((int64_t)blockIdx.x * PG + wave / SLICES) * 64 + lane;
    const bool live = k < N;
    const int64_t kk = live ? k : (N - 1);
    const int64_t ii = order ? (int64_t)order[kk] : kk;
    const float px = pts[3 * ii], py = pts[3 * ii + 1], pz = pts[3 * ii + 2];
    LaneState s;
    // chamfer.py:81-82 transform_points, k-ordered fma chain
    s.p = v3(add_rn(fmaf(M[2], pz, fmaf(M[1], py, mul_rn(M[0], px))), M[3]),
             add_rn(fmaf(M[6], pz, fmaf(M[5], py, mul_rn(M[4], px))), M[7]),
             add_rn(fmaf(M[10], pz, fmaf(M[9], py, mul_rn(M[8], px))), M[11]));
    s.best_f = -1;
    s.best_d2 = INFINITY;
    s.reach = INFINITY;
    s.reach2 = INFINITY;
    int hits = 0;
    scan_mesh<PG, SLICES, false>(m, tile_lds, s, s.p, s.p, hits, ctl);
    merge_slices<PG, SLICES>(tile_lds, s, hits);
    if ((wave % SLICES) != 0) return;
    double acc = 0.0;
    if (live && s.best_f >= 0) {
        const V3 q = closest_on_face(m, m.rec + (int64_t)kRec * m.rec_of_face[s.best_f], s.p);
        const float sd = mul_rn(scale, norm3_unfused(sub(q, s.p)));  // chamfer.py:92
        acc = (double)mul_rn(sd, sd);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) atomicAdd(out_sum + blockIdx.y, acc);
}

// Z-order key of each point inside the box [lo, hi] (device [2][3]): 3 x 10 bits, interleaved.  Sorting queries by it
// makes the 64 points of a wave neighbours in space, which is what the tile culling feeds on.
__global__ __launch_bounds__(256) void morton_keys_kernel(const float* __restrict__ pts, int64_t P,
                                                          const float* __restrict__ box, int* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    unsigned key = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float lo = box[d], hi = box[3 + d];
        float t = (pts[3 * i + d] - lo) / fmaxf(hi - lo, 1e-30f) * 1023.f;
        t = fminf(fmaxf(t, 0.f), 1023.f);  // NaN -> 0
        unsigned c = (unsigned)t;
        c = (c | (c << 16)) & 0x030000FFu;  // spread 10 bits to every third position
        c = (c | (c << 8)) & 0x0300F00Fu;
        c = (c | (c << 4)) & 0x030C30C3u;
        c = (c | (c << 2)) & 0x09249249u;
        key |= c << d;
    }
    keys[i] = (int)key;
}

__global__ void zero_f64_kernel(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// rec_of_face[original id] = position of that face's record (records may be stored in any order)
__global__ void invert_face_order_kernel(const float* __restrict__ rec, int F, int* __restrict__ rec_of_face) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < F) rec_of_face[__float_as_int(rec[(int64_t)kRec * k + 7])] = k;
}

static MeshArgs mesh_args(const pvamd_mesh_t& mesh) {
    MeshArgs m;
    m.normal = mesh.normal;
    m.rec = mesh.rec;
    m.tiles = mesh.tiles;
    m.rec_of_face = mesh.rec_of_face;
    m.F = mesh.F;
    for (int d = 0; d < 3; ++d) m.ray_dir[d] = mesh.ray_dir[d];
    return m;
}

// How many waves share one 64-point group.  The work per point is heavy-tailed once culling is on (a point near the
// medial axis of the mesh is equidistant to much of the surface and must test most triangles exactly), so even when
// there are plenty of points a group is split over 8 waves: it bounds the slowest group's time at 1/8 (measured on
// C5: 45 ms with one wave per group -> see profiles/), at the price of repeating the per-tile bookkeeping per wave.
#ifndef PVAMD_BIG_SLICES
#define PVAMD_BIG_SLICES 8
#endif
static int pick_slices(int64_t point_tiles) {
    return point_tiles >= (int64_t)kNumCU * 16 ? PVAMD_BIG_SLICES : 16;
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_morton_keys(const float* points, int64_t P, const float* box, int32_t* keys_out, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!points || !box || !keys_out) return PVAMD_E_NULL;
    hipLaunchKernelGGL(morton_keys_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, P,
                       box, keys_out);
    return (int)hipGetLastError();
}

extern "C" int pvamd_mesh_prepare(const float* tri, const int32_t* face_id, int32_t F, float abs_margin, float* rec_out,
                                  float* tiles_out, int32_t* rec_of_face_out, void* stream) {
    if (F < 0) return PVAMD_E_SHAPE;
    if (F == 0) return 0;
    if (!tri || !rec_out || !tiles_out || !rec_of_face_out) return PVAMD_E_NULL;
    if (!aligned_to(rec_out, 16)) return PVAMD_E_ALIGN;
    if (!(abs_margin >= 0.f)) return PVAMD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mesh_prepare_records, dim3((F + 255) / 256), dim3(256), 0, s, tri, face_id, F, abs_margin, rec_out);
    hipLaunchKernelGGL(mesh_prepare_tiles, dim3((F + kTile - 1) / kTile), dim3(256), 0, s, rec_out, F, abs_margin, tiles_out);
    hipLaunchKernelGGL(invert_face_order_kernel, dim3((F + 255) / 256), dim3(256), 0, s, rec_out, F, rec_of_face_out);
    return (int)hipGetLastError();
}

extern "C" int pvamd_mesh_query(const pvamd_mesh_t* mesh, const float* points, const int32_t* order, int64_t P,
                                uint64_t jitter_seed, int64_t index_base, float* out_closest, float* out_dist, float* out_grad,
                                int32_t* out_face, float* out_normal, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!mesh || !out_dist || !out_grad) return PVAMD_E_NULL;
    if (mesh->F < 0) return PVAMD_E_SHAPE;
    if (!points || (mesh->F > 0 && (!mesh->rec || !mesh->tiles || !mesh->normal || !mesh->rec_of_face))) return PVAMD_E_NULL;
    const MeshArgs m = mesh_args(*mesh);
    const int64_t ptiles = (P + 63) / 64;
    if (ptiles > 0x7fffffff) return PVAMD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (pick_slices(ptiles)) {
        case 8: hipLaunchKernelGGL((mesh_query_kernel<1, 8>), dim3((unsigned)ptiles), dim3(512), 0, s, m, order, points, P, jitter_seed, index_base, out_closest, out_dist, out_grad, out_face, out_normal); break;
        default: hipLaunchKernelGGL((mesh_query_kernel<1, 16>), dim3((unsigned)ptiles), dim3(1024), 0, s, m, order, points, P, jitter_seed, index_base, out_closest, out_dist, out_grad, out_face, out_normal); break;
    }
    return (int)hipGetLastError();
}

extern "C" int pvamd_chamfer_mesh(const pvamd_mesh_t* mesh, const float* W, int32_t B, const float* points,
                                  const int32_t* order, int64_t N, float scale, double* out_sum, void* stream) {
    if (B < 0 || N < 0) return PVAMD_E_SHAPE;
    if (B == 0) return 0;
    if (!mesh || !out_sum) return PVAMD_E_NULL;
    if (mesh->F < 0) return PVAMD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_f64_kernel, dim3((B + 255) / 256), dim3(256), 0, s, out_sum, B);
    if (N == 0 || mesh->F == 0) return (int)hipGetLastError();
    if (!W || !points || !mesh->rec || !mesh->tiles || !mesh->rec_of_face) return PVAMD_E_NULL;
    const MeshArgs m = mesh_args(*mesh);
    const int64_t ptiles = (N + 63) / 64;
    if (ptiles > 0x7fffffff) return PVAMD_E_SHAPE;
    // y-dimension of a HIP grid is limited to 65535: walk B in slabs
    for (int32_t b0 = 0; b0 < B; b0 += 65535) {
        const int32_t nb = (B - b0) < 65535 ? (B - b0) : 65535;
        switch (pick_slices(ptiles * nb)) {
            case 8: hipLaunchKernelGGL((chamfer_mesh_kernel<1, 8>), dim3((unsigned)ptiles, nb), dim3(512), 0, s, m, order, W + 16 * (int64_t)b0, points, N, scale, out_sum + b0); break;
            default: hipLaunchKernelGGL((chamfer_mesh_kernel<1, 16>), dim3((unsigned)ptiles, nb), dim3(1024), 0, s, m, order, W + 16 * (int64_t)b0, points, N, scale, out_sum + b0); break;
        }
    }
    return (int)hipGetLastError();
}
