// Point x triangle kernels (BASELINE configs C1, C5; also the voxel-cache build and LOOKUP_GT_SDF).
//   mesh_query:   closest surface point + ray-hit parity sign + gradient + face id   (reference sdf.py:122-172,
//                 where it is a device->host copy, two Embree BVH traversals on CPU threads, ~12 numpy passes)
//   chamfer_mesh: per-transform sum of (scale*d)^2 over the transformed points         (reference chamfer.py:79-94)
//
// fp32-VALU bound, not HBM bound.  Structure (details at "The scan" below):
//   * pvamd_mesh_prepare turns the soup into 96-byte records (bounding sphere, in-plane bounding rectangle, corners,
//     original face id) + one bounding sphere per group of 16 and per tile of 256 records.
//   * a block owns 64 points (one per lane) and SLICES waves; a tile of records is staged into LDS once per block and
//     its groups are dealt round-robin to the waves (small P -> many slices so the 1024 SIMDs still fill; very small
//     P -> the tiles of a point group are spread over several blocks as well).
//   * conservative culling, tile -> group -> record sphere -> rectangle: a record is skipped when it provably cannot
//     lower a lane's best d^2 nor be hit by its ray.  The survivors are queued as (record, point) pairs and the exact
//     tests (the oracle's operation sequences) run densely over the queue.  Bit-identical to the plain double loop of
//     oracle/pvamd_oracle.c; with spatially sorted triangles and points this is a flat three-level BVH.
//   * ties in d^2 resolve to the lowest ORIGINAL face id (lexicographic min), independent of processing order.
#include "common.h"
#include "mesh_math.h"
#include "morton.h"

namespace pvamd {

constexpr int kRec = PVAMD_TRI_REC;      // floats per record
constexpr int kTile = PVAMD_TRI_TILE;    // records per tile: 256 * 96 B = 24 KB of LDS
constexpr int kGroup = PVAMD_TRI_GROUP;  // records per group: own bounding sphere
constexpr int kGroupsPerTile = kTile / kGroup;
// record layout (float index):
//   0-2 ctr, 3 r       bounding sphere (ctr = centre of the in-plane bounding rectangle)
//   4-6 u,   7 hu      unit vector along the longest edge, half extent of the triangle along it (about ctr)
//   8-10 v, 11 hv      unit in-plane vector across it, half extent
//  12-14 a, 15 face id (bits)
//  16-18 b, 19 m0      m0: absolute slack of the rectangle test (plane fit + rounding of the frame)
//  20-22 c, 23 0
// `tiles` buffer: [ntiles][4] tile spheres, then [ntiles][16][4] group spheres.

struct MeshArgs {
    const float* normal;
    const float* rec;
    const float* tiles;
    const int* rec_of_face;
    int F;
    double ray_dir[3];
};

// ---------------------------------------------------------------------------------------------------------------
// prepare (double precision throughout; every stored bound is rounded outwards)
// ---------------------------------------------------------------------------------------------------------------
struct D3 { double x, y, z; };
PVAMD_DEV D3 d3(double x, double y, double z) { return D3{x, y, z}; }
PVAMD_DEV D3 dsub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
PVAMD_DEV double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PVAMD_DEV D3 dcross(D3 a, D3 b) { return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PVAMD_DEV D3 dscale(D3 a, double k) { return d3(a.x * k, a.y * k, a.z * k); }
PVAMD_DEV D3 dunit(D3 a, D3 fallback) {
    const double n = sqrt(ddot(a, a));
    return (n > 0.0 && n < 1e300) ? dscale(a, 1.0 / n) : fallback;
}
PVAMD_DEV double dmax3(double a, double b, double c) { return fmax(a, fmax(b, c)); }
PVAMD_DEV double dmin3(double a, double b, double c) { return fmin(a, fmin(b, c)); }

__global__ __launch_bounds__(256) void mesh_prepare_records(const float* __restrict__ tri, const int* __restrict__ face_id,
                                                            int F, float abs_margin, float* __restrict__ rec) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float* t = tri + 9 * (int64_t)f;
    const D3 A = d3(t[0], t[1], t[2]), B = d3(t[3], t[4], t[5]), C = d3(t[6], t[7], t[8]);
    // frame: u along the longest edge, n the plane normal, v = n x u
    const D3 eab = dsub(B, A), ebc = dsub(C, B), eca = dsub(A, C);
    const double lab = ddot(eab, eab), lbc = ddot(ebc, ebc), lca = ddot(eca, eca);
    D3 e = eab;
    if (lbc > lab && lbc >= lca) e = ebc;
    else if (lca > lab && lca > lbc) e = eca;
    const D3 U = dunit(e, d3(1.0, 0.0, 0.0));
    // a vector not parallel to U, for triangles without a usable normal (zero area)
    const double ax = fabs(U.x), ay = fabs(U.y), az = fabs(U.z);
    const D3 axis = (ax <= ay && ax <= az) ? d3(1.0, 0.0, 0.0) : (ay <= az ? d3(0.0, 1.0, 0.0) : d3(0.0, 0.0, 1.0));
    const D3 Nd = dunit(dcross(eab, dsub(C, A)), dunit(dcross(U, axis), d3(0.0, 0.0, 1.0)));
    const D3 V = dunit(dcross(Nd, U), axis);
    // what the kernels will actually use: the frame rounded to fp32
    const float uf[3] = {(float)U.x, (float)U.y, (float)U.z}, vf[3] = {(float)V.x, (float)V.y, (float)V.z};
    const D3 Uf = d3(uf[0], uf[1], uf[2]), Vf = d3(vf[0], vf[1], vf[2]);
    const D3 Nf = dunit(dcross(Uf, Vf), Nd);
    // centre of the bounding rectangle (and of the plane slab) in that frame, then rounded to fp32
    const D3 rb = dsub(B, A), rc = dsub(C, A);
    const double ub = ddot(Uf, rb), uc = ddot(Uf, rc), vb = ddot(Vf, rb), vc = ddot(Vf, rc), nb = ddot(Nf, rb), nc = ddot(Nf, rc);
    const double um = 0.5 * (dmin3(0.0, ub, uc) + dmax3(0.0, ub, uc)), vm = 0.5 * (dmin3(0.0, vb, vc) + dmax3(0.0, vb, vc)),
                 nm = 0.5 * (dmin3(0.0, nb, nc) + dmax3(0.0, nb, nc));
    const float cf[3] = {(float)(A.x + Uf.x * um + Vf.x * vm + Nf.x * nm), (float)(A.y + Uf.y * um + Vf.y * vm + Nf.y * nm),
                         (float)(A.z + Uf.z * um + Vf.z * vm + Nf.z * nm)};
    const D3 Cf = d3(cf[0], cf[1], cf[2]);
    // extents about the ROUNDED centre in the ROUNDED frame: valid for exactly the numbers the kernels see
    const D3 qa = dsub(A, Cf), qb = dsub(B, Cf), qc = dsub(C, Cf);
    const double hu = dmax3(fabs(ddot(Uf, qa)), fabs(ddot(Uf, qb)), fabs(ddot(Uf, qc)));
    const double hv = dmax3(fabs(ddot(Vf, qa)), fabs(ddot(Vf, qb)), fabs(ddot(Vf, qc)));
    const double sl = dmax3(fabs(ddot(Nf, qa)), fabs(ddot(Nf, qb)), fabs(ddot(Nf, qc)));
    const double r = sqrt(dmax3(ddot(qa, qa), ddot(qb, qb), ddot(qc, qc)));
    float* o = rec + (int64_t)kRec * f;
    o[0] = cf[0]; o[1] = cf[1]; o[2] = cf[2]; o[3] = (float)(r * 1.00001) + abs_margin;
    o[4] = uf[0]; o[5] = uf[1]; o[6] = uf[2]; o[7] = (float)(hu * 1.00001) + abs_margin;
    o[8] = vf[0]; o[9] = vf[1]; o[10] = vf[2]; o[11] = (float)(hv * 1.00001) + abs_margin;
    o[12] = t[0]; o[13] = t[1]; o[14] = t[2]; o[15] = __int_as_float(face_id ? face_id[f] : f);
    o[16] = t[3]; o[17] = t[4]; o[18] = t[5]; o[19] = (float)(1.01e6 * sl * sl) + abs_margin * abs_margin;
    o[20] = t[6]; o[21] = t[7]; o[22] = t[8]; o[23] = 0.f;
}

// sphere around the mean of `width` consecutive lanes' record centres, radius = max(|c_i - mean| + r_i), inflated
PVAMD_DEV void enclose(bool live, float cx, float cy, float cz, float r, int width, float abs_margin, float out[4]) {
    float sx = live ? cx : 0.f, sy = live ? cy : 0.f, sz = live ? cz : 0.f, sn = live ? 1.f : 0.f;
    for (int off = width / 2; off > 0; off >>= 1) {
        sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64);
        sz += __shfl_xor(sz, off, 64); sn += __shfl_xor(sn, off, 64);
    }
    sn = fmaxf(sn, 1.f);
    const float mx = sx / sn, my = sy / sn, mz = sz / sn;
    float reach = 0.f;
    if (live) {
        const float dx = cx - mx, dy = cy - my, dz = cz - mz;
        reach = sqrt_rn(dx * dx + dy * dy + dz * dz) * 1.00001f + r;
    }
    for (int off = width / 2; off > 0; off >>= 1) reach = fmaxf(reach, __shfl_xor(reach, off, 64));
    out[0] = mx; out[1] = my; out[2] = mz;
    out[3] = reach * 1.00001f + abs_margin;
}

// one block per tile: the tile sphere and its 16 group spheres
__global__ __launch_bounds__(256) void mesh_prepare_tiles(const float* __restrict__ rec, int F, float abs_margin,
                                                          float* __restrict__ tiles) {
    __shared__ float sh[4][4];
    const int tile = blockIdx.x, ntiles = gridDim.x;
    const int f = tile * kTile + threadIdx.x;
    const bool live = f < F;
    const float* o = rec + (int64_t)kRec * (live ? f : (F - 1));
    const float cx = o[0], cy = o[1], cz = o[2], r = o[3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float g[4];
    enclose(live, cx, cy, cz, r, kGroup, abs_margin, g);
    if (live && (threadIdx.x % kGroup) == 0) {
        float* w = tiles + 4 * (int64_t)ntiles + 4 * ((int64_t)tile * kGroupsPerTile + threadIdx.x / kGroup);
        w[0] = g[0]; w[1] = g[1]; w[2] = g[2]; w[3] = g[3];
    }
    // the tile: same construction over all of its records
    float sx = live ? cx : 0.f, sy = live ? cy : 0.f, sz = live ? cz : 0.f, sn = live ? 1.f : 0.f;
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64);
        sz += __shfl_xor(sz, off, 64); sn += __shfl_xor(sn, off, 64);
    }
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
        reach = sqrt_rn(dx * dx + dy * dy + dz * dz) * 1.00001f + r;
    }
    for (int off = 32; off > 0; off >>= 1) reach = fmaxf(reach, __shfl_xor(reach, off, 64));
    if (lane == 0) sh[wave][0] = reach;
    __syncthreads();
    if (threadIdx.x == 0) {
        float* w = tiles + 4 * (int64_t)tile;
        w[0] = mx; w[1] = my; w[2] = mz;
        w[3] = fmaxf(fmaxf(sh[0][0], sh[1][0]), fmaxf(sh[2][0], sh[3][0])) * 1.00001f + abs_margin;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// query
// ---------------------------------------------------------------------------------------------------------------
// np.linalg.norm of a float32 3-vector (sdf.py:141): products and sums rounded separately, left to right
PVAMD_DEV float norm3_unfused(V3 g) {
    return sqrt_rn(add_rn(add_rn(mul_rn(g.x, g.x), mul_rn(g.y, g.y)), mul_rn(g.z, g.z)));
}

struct LaneState {
    V3 p;
    float reach;             // inflated upper bound on the distance to the mesh: "closer than this could still win"
    float reach2m, reach2x;  // reach^2 * 1.00001 and 2 * reach * 1.00001: the reach terms of the sphere test
};

PVAMD_DEV void set_reach(LaneState& s, float reach) {
    s.reach = reach;
    s.reach2m = reach * reach * 1.00001f;
    s.reach2x = 2.00002f * reach;
}

// sphere (ctr, r) cannot contain a point closer to p than the current best:  |p-ctr|^2 > (r + reach)^2 * 1.00001
PVAMD_DEV bool sphere_may_improve(const LaneState& s, float dist2, float r) {
    return !(dist2 > fmaf(r, fmaf(r, 1.00001f, s.reach2x), s.reach2m));
}

// sphere (ctr, r) may be crossed by the ray p + t*dn, t > 0 (dn unit):  distance from ctr to the line <= r, not behind p
PVAMD_DEV bool sphere_may_hit(float dist2, float tp, float r) {
    const float perp2 = fmaf(-tp, tp, dist2);
    return !(perp2 > fmaf(r, r, 2e-6f * dist2)) && !(tp < -r);
}

// Second, tighter filter for the closest-point pairs that pass the sphere test.  The triangle lies inside the rectangle
// |u.(x-ctr)| <= hu, |v.(x-ctr)| <= hv of (nearly) its own plane, so with w = ctr - p
//     |x - p|^2  >=  max(0, |w|^2 - (u.w)^2 - (v.w)^2)  +  max(0, |u.w| - hu)^2  +  max(0, |v.w| - hv)^2  -  slack,
// slack = 8e-6 |w|^2 + m0 covering the fp32 evaluation, the rounded frame (not exactly orthonormal) and the distance of
// the corners from the frame's plane (m0 = 1e6 s^2: 2|w|s <= 1e-6 |w|^2 + 1e6 s^2); hu, hv and m0 were computed in
// float64 for exactly the rounded ctr/u/v stored here (mesh_prepare_records).  Unlike the sphere this stays tight for
// long, thin and large triangles.  Any NaN/inf makes the comparison false: the pair is kept.
PVAMD_DEV bool rect_may_improve(const LaneState& s, V3 w, float dist2, const float* __restrict__ o) {
    const float u = dot(v3(o[4], o[5], o[6]), w), v = dot(v3(o[8], o[9], o[10]), w);
    const float h2 = fmaxf(fmaf(-v, v, fmaf(-u, u, dist2)), 0.f);
    const float du = fmaxf(fabsf(u) - o[7], 0.f), dv = fmaxf(fabsf(v) - o[11], 0.f);
    const float lb2 = fmaf(dv, dv, fmaf(du, du, h2));
    return !(lb2 > fmaf(8e-6f, dist2, s.reach2m + o[19]));
}

#ifdef PVAMD_MESH_STATS
__device__ unsigned long long g_stats[16];
#define STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
extern "C" int pvamd_debug_stats(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(g_stats));
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)); }
    return 0;
}
#else
#define STAT(i, v)
#endif

// ---------------------------------------------------------------------------------------------------------------
// The scan.  One block = 64 points (one per lane) x SLICES waves that all hold the same 64 points.
//   seed    every wave bounds the distance to the mesh from its share of the tile spheres (min over tiles of
//           |p - ctr| + r: each sphere contains a whole triangle); the merged bound seeds `reach`, and the tiles
//           nearest to lane 0 / lane 63 (the two ends of a Morton-sorted run of points) are visited first, which
//           tightens reach to about the true distance before anything else is looked at.
//   vote    one pass over the remaining tile spheres, split across the waves, flags the tiles some lane still needs
//           (closer than reach, or crossed by the lane's ray) in an LDS bit mask; unflagged tiles cost nothing more.
//   visit   a flagged tile is staged into LDS by the whole block (2 barriers); groups of 16 records are dealt
//           round-robin to the waves; group sphere, then record spheres, wave-uniform (LDS broadcast reads).
//   narrow  the broad phase only QUEUES (record, point) pairs -- per wave, in LDS, one queue for closest-point pairs
//           and one for ray pairs.  Whenever 64 are waiting the wave runs them densely: lane i takes pair i, whichever
//           point and record that is (typically ~10-60 % of the lanes of a wave need a given record, so running the
//           exact test per record would idle the rest).  Results fold into per-point LDS slots shared by all waves:
//           a 64-bit atomicMin on (d2 bits << 32 | face id) IS the lexicographic "smallest d2, lowest original face
//           id" rule (d2 >= +0, so its bit pattern orders like the float; a NaN sorts above +inf and never wins), hit
//           counts by atomicAdd.  After a drain every lane tightens its reach from the shared slot, so the waves help
//           each other cull.
// A skipped record provably cannot lower a lane's best d^2 nor be hit by its ray, min and + are order-free, so the
// results are bit-identical to the plain double loop of oracle/pvamd_oracle.c.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kQueueCap = 128;     // entries per queue per wave (uint16: record << 6 | owner lane); < 64 + 64 in use
constexpr int kVoteTiles = 2048;   // tiles per vote pass (64 mask words)
constexpr unsigned long long kBestInit = 0x7F80000000000000ull;  // (+inf, face 0): a finite candidate always wins

struct MeshShared {
    float tile[kTile * kRec];  // doubles as [SLICES][64] u64 scratch for the seed merge before the first tile
    float group[kGroupsPerTile * 4];
    unsigned long long best[64];
    int hits[64];
    float pt[64 * 3];
    float dir[64 * 3];  // jittered ray direction (exact test)
    float dn[64 * 3];   // its unit vector (sphere tests)
    unsigned mask[kVoteTiles / 32];
};

// enqueue the lanes of `mask` for record j (wave-uniform j): the k-th set lane writes slot n + k
PVAMD_DEV void enqueue(unsigned short* q, int& n, unsigned long long mask, bool mine, int j) {
    const int lane = threadIdx.x & 63;
    if (mine) q[n + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)((j << 6) | lane);
    n += __popcll(mask);
}

// Run fn(entry) densely over the queue: always the first (up to) 64 entries; the remainder (n - 64 < 64 by
// construction) too when `everything`, else it moves to the front and waits for more company.
template <class Fn>
PVAMD_DEV void drain(unsigned short* q, int& n, bool everything, Fn&& fn) {
    const int lane = threadIdx.x & 63;
    PVAMD_WAVE_SYNC();  // entries were written by other lanes
    STAT(6, 1);
    if (lane < n) fn((unsigned)q[lane]);
    if (n > 64) {
        const int rem = n - 64;
        const unsigned v = lane < rem ? (unsigned)q[64 + lane] : 0u;
        if (everything) {
            if (lane < rem) fn(v);
            n = 0;
        } else {
            if (lane < rem) q[lane] = (unsigned short)v;
            n = rem;
        }
    } else {
        n = 0;
    }
    PVAMD_WAVE_SYNC();
}

PVAMD_DEV void drain_closest(MeshShared& sh, unsigned short* q, int& n, bool everything, LaneState& s) {
    if (n == 0) return;
    drain(q, n, everything, [&](unsigned e) {
        const float* o = sh.tile + kRec * (e >> 6);
        const int owner = e & 63;
        const V3 p = v3(sh.pt[3 * owner], sh.pt[3 * owner + 1], sh.pt[3 * owner + 2]);
        const V3 c = closest_point_triangle(p, v3(o[12], o[13], o[14]), v3(o[16], o[17], o[18]), v3(o[20], o[21], o[22]));
        const V3 g = sub(c, p);
        const float d2 = dot(g, g);
        atomicMin(&sh.best[owner],
                  ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(o[15]));
    });
    // tighten this lane's reach from the block-shared slot (other waves' finds included)
    const float d2 = __int_as_float((int)(unsigned)(sh.best[threadIdx.x & 63] >> 32));
    const float reach = sqrt_rn(d2) * 1.00001f;
    if (reach < s.reach) set_reach(s, reach);
}

PVAMD_DEV void drain_rays(MeshShared& sh, unsigned short* q, int& n, bool everything) {
    if (n == 0) return;
    drain(q, n, everything, [&](unsigned e) {
        const float* o = sh.tile + kRec * (e >> 6);
        const int owner = e & 63;
        const V3 p = v3(sh.pt[3 * owner], sh.pt[3 * owner + 1], sh.pt[3 * owner + 2]);
        const V3 d = v3(sh.dir[3 * owner], sh.dir[3 * owner + 1], sh.dir[3 * owner + 2]);
        if (ray_hits_triangle(p, d, v3(o[12], o[13], o[14]), v3(o[16], o[17], o[18]), v3(o[20], o[21], o[22])))
            atomicAdd(&sh.hits[owner], 1);
    });
}

// does any lane of this wave still need tile ti?  (ti wave-uniform: scalar loads)
template <bool WITH_RAY>
PVAMD_DEV bool wave_needs_tile(const MeshArgs& m, int ti, const LaneState& s, V3 dn) {
    const float* ts = m.tiles + 4 * (int64_t)ti;
    const V3 wt = v3(ts[0] - s.p.x, ts[1] - s.p.y, ts[2] - s.p.z);
    const float tdist2 = dot(wt, wt);
    bool need = sphere_may_improve(s, tdist2, ts[3]);
    if (WITH_RAY) need = need || sphere_may_hit(tdist2, dot(wt, dn), ts[3]);
    return __any(need);
}

// Stage tile ti into LDS and run this wave's share of it.  Entry: nobody still reads the previous tile.  Exit: ditto.
template <int SLICES, bool WITH_RAY>
PVAMD_DEV void visit_tile(const MeshArgs& m, MeshShared& sh, int ti, int wave, unsigned short* qc, unsigned short* qr,
                          LaneState& s, V3 dn) {
    STAT(1, wave == 0);
    const int n = min(kTile, m.F - ti * kTile);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(m.rec + (int64_t)kRec * kTile * ti);
        f32x4_alias* dst = reinterpret_cast<f32x4_alias*>(sh.tile);
        for (int k = threadIdx.x; k < n * (kRec / 4); k += 64 * SLICES) dst[k] = src[k];
        if (threadIdx.x < kGroupsPerTile * 4)
            sh.group[threadIdx.x] = m.tiles[4 * (int64_t)((m.F + kTile - 1) / kTile) + (int64_t)ti * kGroupsPerTile * 4 + threadIdx.x];
    }
    __syncthreads();
    if (wave_needs_tile<WITH_RAY>(m, ti, s, dn)) {  // reach may have tightened since the vote
        STAT(2, 1);
        int nc = 0, nr = 0;
        for (int g0 = wave * kGroup; g0 < n; g0 += SLICES * kGroup) {
            const float* og = sh.group + 4 * (g0 / kGroup);  // wave-uniform addresses below: LDS broadcast reads
            const V3 wg = v3(og[0] - s.p.x, og[1] - s.p.y, og[2] - s.p.z);
            const float rg = og[3];
            const float gdist2 = dot(wg, wg);
            bool gneed = sphere_may_improve(s, gdist2, rg);
            if (WITH_RAY) gneed = gneed || sphere_may_hit(gdist2, dot(wg, dn), rg);
            STAT(9, 1);
            if (!__any(gneed)) continue;
            const int g1 = min(g0 + kGroup, n);
            for (int j = g0; j < g1; ++j) {
                const float* o = sh.tile + kRec * j;
                const V3 w = v3(o[0] - s.p.x, o[1] - s.p.y, o[2] - s.p.z);
                const float r = o[3];
                const float dist2 = dot(w, w);
                const bool near_c = sphere_may_improve(s, dist2, r);
                STAT(3, 1);
                if (__any(near_c)) {
                    STAT(10, 1);
                    const bool need_c = near_c && rect_may_improve(s, w, dist2, o);
                    const unsigned long long mc = __ballot(need_c);
                    if (mc) {
                        STAT(4, __popcll(mc)); STAT(7, 1);
                        enqueue(qc, nc, mc, need_c, j);
                        if (nc >= 64) drain_closest(sh, qc, nc, false, s);
                    }
                }
                if (WITH_RAY) {
                    const bool need_r = sphere_may_hit(dist2, dot(w, dn), r);
                    const unsigned long long mr = __ballot(need_r);
                    if (mr) {
                        STAT(5, __popcll(mr)); STAT(8, 1);
                        enqueue(qr, nr, mr, need_r, j);
                        if (nr >= 64) drain_rays(sh, qr, nr, false);
                    }
                }
            }
        }
        // the tile is about to be replaced: finish everything that points into it
        drain_closest(sh, qc, nc, true, s);
        if (WITH_RAY) drain_rays(sh, qr, nr, true);
    }
    __syncthreads();
}

// seed (see above): sets s.reach, returns the two tiles to visit first.  On entry: s.p set by every wave; wave 0 has
// filled sh.pt / sh.dir / sh.dn (no barrier needed yet).  Initialises sh.best / sh.hits; two barriers inside.
template <int SLICES, bool WITH_RAY>
PVAMD_DEV void scan_seed(const MeshArgs& m, MeshShared& sh, LaneState& s, V3& dn, int& first0, int& first1) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntiles = (m.F + kTile - 1) / kTile;
    {
        float bound = INFINITY;
        int nearest = 0;
        for (int ti = wave; ti < ntiles; ti += SLICES) {
            const float* ts = m.tiles + 4 * (int64_t)ti;
            const V3 w = v3(ts[0] - s.p.x, ts[1] - s.p.y, ts[2] - s.p.z);
            const float b = fast_sqrt(dot(w, w)) * 1.00001f + (ts[3] + 1.1e-19f);  // slack: 1 ulp + the flushed denormals
            if (b < bound) {  // a NaN point never passes: keeps INFINITY / tile 0
                bound = b;
                nearest = ti;
            }
        }
        unsigned long long* scratch = reinterpret_cast<unsigned long long*>(sh.tile);
        scratch[wave * 64 + lane] = ((unsigned long long)(unsigned)__float_as_int(bound) << 32) | (unsigned)nearest;
        if (wave == 0) {
            sh.best[lane] = kBestInit;
            sh.hits[lane] = 0;
        }
    }
    __syncthreads();
    {
        const unsigned long long* scratch = reinterpret_cast<const unsigned long long*>(sh.tile);
        unsigned long long lo = scratch[lane];
#pragma unroll
        for (int w = 1; w < SLICES; ++w) {
            const unsigned long long v = scratch[w * 64 + lane];
            lo = v < lo ? v : lo;  // bound >= 0: bit order = float order; ties -> lowest tile index
        }
        set_reach(s, __int_as_float((int)(unsigned)(lo >> 32)) * 1.00001f);
        first0 = __shfl((int)(unsigned)lo, 0, 64);
        first1 = __shfl((int)(unsigned)lo, 63, 64);
        if (WITH_RAY) dn = v3(sh.dn[3 * lane], sh.dn[3 * lane + 1], sh.dn[3 * lane + 2]);
    }
    __syncthreads();  // scratch consumed: the tile buffer may be overwritten
}

// vote + visit over the tiles ti with ti % nparts == part, first0 / first1 excepted (they were visited before)
template <int SLICES, bool WITH_RAY>
PVAMD_DEV void scan_rest(const MeshArgs& m, MeshShared& sh, unsigned short* qc, unsigned short* qr, LaneState& s, V3 dn,
                         int first0, int first1, int part, int nparts) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntiles = (m.F + kTile - 1) / kTile;
    for (int base = 0; base < ntiles; base += kVoteTiles) {
        const int nwords = (min(kVoteTiles, ntiles - base) + 31) / 32;
        if (threadIdx.x < kVoteTiles / 32) sh.mask[threadIdx.x] = 0u;
        __syncthreads();
        for (int word = 0; word < nwords; ++word) {
            unsigned bits = 0u;
            for (int t = wave; t < 32; t += SLICES) {
                const int ti = base + 32 * word + t;
                if (ti >= ntiles || ti == first0 || ti == first1) continue;
                if (nparts > 1 && (ti % nparts) != part) continue;
                STAT(0, 1);
                if (wave_needs_tile<WITH_RAY>(m, ti, s, dn)) bits |= 1u << t;
            }
            if (bits != 0u && lane == 0) atomicOr(&sh.mask[word], bits);
        }
        __syncthreads();
        for (int word = 0; word < nwords; ++word) {
            unsigned todo = __builtin_amdgcn_readfirstlane(sh.mask[word]);
            while (todo != 0u) {
                const int t = __ffs(todo) - 1;
                todo &= todo - 1u;
                visit_tile<SLICES, WITH_RAY>(m, sh, base + 32 * word + t, wave, qc, qr, s, dn);
            }
        }
        if (base + kVoteTiles < ntiles) __syncthreads();  // every wave has read the mask before it is cleared again
    }
}

// The whole scan in one block.  On exit (after a barrier): sh.best[lane] / sh.hits[lane] = result for point `lane`.
template <int SLICES, bool WITH_RAY>
PVAMD_DEV void scan_mesh(const MeshArgs& m, MeshShared& sh, unsigned short* qc, unsigned short* qr, LaneState& s) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    V3 dn = s.p;
    int first0, first1;
    scan_seed<SLICES, WITH_RAY>(m, sh, s, dn, first0, first1);
    if (m.F <= 0) return;
    visit_tile<SLICES, WITH_RAY>(m, sh, first0, wave, qc, qr, s, dn);
    if (first1 != first0) visit_tile<SLICES, WITH_RAY>(m, sh, first1, wave, qc, qr, s, dn);
    scan_rest<SLICES, WITH_RAY>(m, sh, qc, qr, s, dn, first0, first1, 0, 1);
}

// the closest point on the winning face, recomputed from its corners (same operations as during the scan)
PVAMD_DEV V3 closest_on_face(const MeshArgs& m, const float* __restrict__ tri_of_face, V3 p) {
    const float* o = tri_of_face;
    return closest_point_triangle(p, v3(o[12], o[13], o[14]), v3(o[16], o[17], o[18]), v3(o[20], o[21], o[22]));
}

struct QueryOut {
    float* closest;
    float* dist;
    float* grad;
    int* face;
    float* normal;
};

// the point this lane owns and (wave 0) the per-point LDS tables incl. the jittered ray
PVAMD_DEV int64_t load_query_point(const MeshArgs& m, MeshShared& sh, const int* __restrict__ order,
                                   const float* __restrict__ pts, int64_t P, uint64_t seed, int64_t index_base, bool with_ray,
                                   LaneState& s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;  // position in processing order
    const int64_t kk = k < P ? k : (P - 1);
    const int64_t i = order ? (int64_t)order[kk] : kk;  // spatially sorted processing; outputs stay in caller order
    s.p = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    set_reach(s, INFINITY);
    if (wave == 0) {  // the other waves read these from LDS
        sh.pt[3 * lane] = s.p.x; sh.pt[3 * lane + 1] = s.p.y; sh.pt[3 * lane + 2] = s.p.z;
        if (with_ray) {
            const V3 dir = jitter_dir(m.ray_dir, seed, index_base + i);
            const float inv_len = 1.f / sqrt_rn(dot(dir, dir));
            sh.dir[3 * lane] = dir.x; sh.dir[3 * lane + 1] = dir.y; sh.dir[3 * lane + 2] = dir.z;
            sh.dn[3 * lane] = dir.x * inv_len; sh.dn[3 * lane + 1] = dir.y * inv_len; sh.dn[3 * lane + 2] = dir.z * inv_len;
        }
    }
    return i;
}

// sdf.py:139-171 from the winning (d2, face) and the hit count
PVAMD_DEV void write_query(const MeshArgs& m, const QueryOut& out, int64_t i, V3 p, unsigned long long found, int hits) {
    const int f = (unsigned)(found >> 32) == 0x7F800000u ? -1 : (int)(unsigned)found;
    V3 q = v3(NAN, NAN, NAN);
    if (f >= 0) q = closest_on_face(m, m.rec + (int64_t)kRec * m.rec_of_face[f], p);
    V3 g = sub(q, p);                            // sdf.py:139
    float d = norm3_unfused(g);                  // :141
    if (d > 0.f) g = v3(div_rn(g.x, d), div_rn(g.y, d), div_rn(g.z, d));  // :143-144
    if (hits & 1) d = -d;                        // :154-155 inside: negative distance
    else g = v3(-g.x, -g.y, -g.z);               // :157 outside: point away from the surface
    if (fabsf(d) < 1e-3f && f >= 0) {            // :162-164 on the surface: use the face normal
        g = v3(m.normal[3 * f], m.normal[3 * f + 1], m.normal[3 * f + 2]);
    }
    if (out.closest) {
        out.closest[3 * i] = q.x;
        out.closest[3 * i + 1] = q.y;
        out.closest[3 * i + 2] = q.z;
    }
    out.dist[i] = d;
    out.grad[3 * i] = g.x;
    out.grad[3 * i + 1] = g.y;
    out.grad[3 * i + 2] = g.z;
    if (out.face) out.face[i] = f;
    if (out.normal) {                            // :169-171
        out.normal[3 * i] = f >= 0 ? m.normal[3 * f] : NAN;
        out.normal[3 * i + 1] = f >= 0 ? m.normal[3 * f + 1] : NAN;
        out.normal[3 * i + 2] = f >= 0 ? m.normal[3 * f + 2] : NAN;
    }
}

template <int SLICES>
__global__ __launch_bounds__(64 * SLICES) void mesh_query_kernel(MeshArgs m, const int* __restrict__ order,
                                                                const float* __restrict__ pts, int64_t P,
                                                                uint64_t seed, int64_t index_base, QueryOut out) {
    __shared__ __attribute__((aligned(16))) MeshShared sh;
    __shared__ unsigned short queue_c[SLICES][kQueueCap], queue_r[SLICES][kQueueCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LaneState s;
    const int64_t i = load_query_point(m, sh, order, pts, P, seed, index_base, true, s);
    scan_mesh<SLICES, true>(m, sh, queue_c[wave], queue_r[wave], s);
    if (wave != 0 || (int64_t)blockIdx.x * 64 + lane >= P) return;
    write_query(m, out, i, s.p, sh.best[lane], sh.hits[lane]);
}

// ---- few points: the tiles of one point group are spread over several blocks --------------------------------
// A block of 64 points walks its flagged tiles one after the other; with only a few hundred blocks in flight that
// serial walk, not throughput, sets the time.  Three launches instead:
//   first   (one block per group)      seed + the two nearest tiles -> (d2, face), hit count, first0/first1 to scratch
//   rest    (nparts blocks per group)  start from the scratch values, vote + visit the tiles ti % nparts == part,
//                                      fold into scratch with global atomicMin / atomicAdd
//   finish  (one wave per group)       outputs from the scratch values
// scratch: u64 best[G*64], int hits[G*64], int firsts[G*2], G = ceil(P/64) groups, indexed by processing position.
struct SplitScratch {
    unsigned long long* best;
    int* hits;
    int* firsts;
};
PVAMD_DEV SplitScratch split_scratch(void* scratch, int64_t groups) {
    SplitScratch r;
    r.best = reinterpret_cast<unsigned long long*>(scratch);
    r.hits = reinterpret_cast<int*>(r.best + groups * 64);
    r.firsts = r.hits + groups * 64;
    return r;
}

template <int SLICES>
__global__ __launch_bounds__(64 * SLICES) void mesh_query_first_kernel(MeshArgs m, const int* __restrict__ order,
                                                                      const float* __restrict__ pts, int64_t P,
                                                                      uint64_t seed, int64_t index_base, void* scratch) {
    __shared__ __attribute__((aligned(16))) MeshShared sh;
    __shared__ unsigned short queue_c[SLICES][kQueueCap], queue_r[SLICES][kQueueCap];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LaneState s;
    load_query_point(m, sh, order, pts, P, seed, index_base, true, s);
    V3 dn = s.p;
    int first0, first1;
    scan_seed<SLICES, true>(m, sh, s, dn, first0, first1);
    visit_tile<SLICES, true>(m, sh, first0, wave, queue_c[wave], queue_r[wave], s, dn);
    if (first1 != first0) visit_tile<SLICES, true>(m, sh, first1, wave, queue_c[wave], queue_r[wave], s, dn);
    if (wave != 0) return;
    const SplitScratch sc = split_scratch(scratch, gridDim.x);
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    sc.best[k] = sh.best[lane];
    sc.hits[k] = sh.hits[lane];
    if (lane == 0) {
        sc.firsts[2 * blockIdx.x] = first0;
        sc.firsts[2 * blockIdx.x + 1] = first1;
    }
}

template <int SLICES>
__global__ __launch_bounds__(64 * SLICES) void mesh_query_rest_kernel(MeshArgs m, const int* __restrict__ order,
                                                                     const float* __restrict__ pts, int64_t P,
                                                                     uint64_t seed, int64_t index_base, void* scratch) {
    __shared__ __attribute__((aligned(16))) MeshShared sh;
    __shared__ unsigned short queue_c[SLICES][kQueueCap], queue_r[SLICES][kQueueCap];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LaneState s;
    load_query_point(m, sh, order, pts, P, seed, index_base, true, s);
    const SplitScratch sc = split_scratch(scratch, gridDim.x);
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const unsigned long long start = sc.best[k];
    if (wave == 0) {
        sh.best[lane] = start;
        sh.hits[lane] = 0;
    }
    {
        const float reach = sqrt_rn(__int_as_float((int)(unsigned)(start >> 32))) * 1.00001f;  // inf while nothing found
        set_reach(s, reach);
    }
    __syncthreads();
    const V3 dn = v3(sh.dn[3 * lane], sh.dn[3 * lane + 1], sh.dn[3 * lane + 2]);
    scan_rest<SLICES, true>(m, sh, queue_c[wave], queue_r[wave], s, dn, sc.firsts[2 * blockIdx.x], sc.firsts[2 * blockIdx.x + 1],
                            (int)blockIdx.y, (int)gridDim.y);
    __syncthreads();
    if (wave != 0) return;
    if (sh.best[lane] < start) atomicMin(&sc.best[k], sh.best[lane]);
    if (sh.hits[lane] != 0) atomicAdd(&sc.hits[k], sh.hits[lane]);
}

__global__ __launch_bounds__(64) void mesh_query_finish_kernel(MeshArgs m, const int* __restrict__ order,
                                                               const float* __restrict__ pts, int64_t P, const void* scratch,
                                                               QueryOut out) {
    const int64_t k = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (k >= P) return;
    const SplitScratch sc = split_scratch(const_cast<void*>(scratch), gridDim.x);
    const int64_t i = order ? (int64_t)order[k] : k;
    write_query(m, out, i, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), sc.best[k], sc.hits[k]);
}

// grid: x = tiles of 64 points, y = transform b
template <int SLICES>
__global__ __launch_bounds__(64 * SLICES) void chamfer_mesh_kernel(MeshArgs m, const int* __restrict__ order,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ pts, int64_t N, float scale,
                                                                  double* __restrict__ out_sum) {
    __shared__ __attribute__((aligned(16))) MeshShared sh;
    __shared__ unsigned short queue_c[SLICES][kQueueCap];
    const float* M = W + 16 * (int64_t)blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const bool live = k < N;
    const int64_t kk = live ? k : (N - 1);
    const int64_t ii = order ? (int64_t)order[kk] : kk;
    const float px = pts[3 * ii], py = pts[3 * ii + 1], pz = pts[3 * ii + 2];
    LaneState s;
    // chamfer.py:81-82 transform_points, k-ordered fma chain
    s.p = v3(add_rn(fmaf(M[2], pz, fmaf(M[1], py, mul_rn(M[0], px))), M[3]),
             add_rn(fmaf(M[6], pz, fmaf(M[5], py, mul_rn(M[4], px))), M[7]),
             add_rn(fmaf(M[10], pz, fmaf(M[9], py, mul_rn(M[8], px))), M[11]));
    set_reach(s, INFINITY);
    if (wave == 0) { sh.pt[3 * lane] = s.p.x; sh.pt[3 * lane + 1] = s.p.y; sh.pt[3 * lane + 2] = s.p.z; }
    scan_mesh<SLICES, false>(m, sh, queue_c[wave], queue_c[wave], s);
    if (wave != 0) return;
    const unsigned long long found = sh.best[lane];
    const int best_f = (unsigned)(found >> 32) == 0x7F800000u ? -1 : (int)(unsigned)found;
    double acc = 0.0;
    if (live && best_f >= 0) {
        const V3 q = closest_on_face(m, m.rec + (int64_t)kRec * m.rec_of_face[best_f], s.p);
        const float sd = mul_rn(scale, norm3_unfused(sub(q, s.p)));  // chamfer.py:92
        acc = (double)mul_rn(sd, sd);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) atomicAdd(out_sum + blockIdx.y, acc);
}

// Z-order key of each point inside the box [lo, hi] (device [2][3]).  Sorting queries by it makes the 64 points of a
// wave neighbours in space, which is what the tile culling feeds on.
__global__ __launch_bounds__(256) void morton_keys_kernel(const float* __restrict__ pts, int64_t P,
                                                          const float* __restrict__ box, int* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float lo[3] = {box[0], box[1], box[2]}, hi[3] = {box[3], box[4], box[5]};
    keys[i] = (int)morton_key30(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], lo, hi);
}

// axis-aligned bounds of the finite coordinates of a point set, for the Morton keys (order_code: morton.h); box holds
// the codes until aabb_decode_kernel.
__global__ void aabb_init_kernel(unsigned* box) {
    if (threadIdx.x < 3) box[threadIdx.x] = order_code(INFINITY);
    else if (threadIdx.x < 6) box[threadIdx.x] = order_code(-INFINITY);
}
__global__ __launch_bounds__(256) void aabb_reduce_kernel(const float* __restrict__ pts, int64_t P, unsigned* box) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = pts[3 * i + d];
            if (fabsf(v) < INFINITY) {  // false for NaN and +-inf
                lo[d] = fminf(lo[d], v);
                hi[d] = fmaxf(hi[d], v);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
    }
    // one set of atomics per block: thousands of waves hammering the same six words serialise in L2 (measured: 0.56 ms
    // for 2 M points with one set per wave, against ~10 us of streaming)
    __shared__ float part[4][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { part[wave][d] = lo[d]; part[wave][3 + d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        atomicMin(box + d, order_code(fminf(fminf(part[0][d], part[1][d]), fminf(part[2][d], part[3][d]))));
    } else if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        atomicMax(box + d, order_code(fmaxf(fmaxf(part[0][d], part[1][d]), fmaxf(part[2][d], part[3][d]))));
    }
}
__global__ void aabb_decode_kernel(unsigned* box) {
    if (threadIdx.x < 6) reinterpret_cast<float*>(box)[threadIdx.x] = order_decode(box[threadIdx.x]);
}

__global__ void zero_f64_kernel(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// rec_of_face[original id] = position of that face's record (records may be stored in any order)
__global__ void invert_face_order_kernel(const float* __restrict__ rec, int F, int* __restrict__ rec_of_face) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < F) rec_of_face[__float_as_int(rec[(int64_t)kRec * k + 15])] = k;
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

// How many waves share one 64-point group.
//   few point groups      -> 16 (then 8), so that the 1024 SIMDs still fill;
//   many groups           -> 4: the least replicated point-side work (A/B at 2 M points on the 62-tile drill: grid 5.4 ->
//                            3.3 ms, random box 2.7 -> 1.9 ms, near-surface chamfer 2.0 -> 1.6 ms against 8; 2 is slower);
//   many groups AND tiles -> 8: the work per group is heavy-tailed (a point near the medial axis is equidistant to much
//                            of the surface and walks most tiles), and on a mesh of hundreds of tiles the slowest
//                            groups, not the throughput, set the kernel time (C5, 389 tiles: 8.2 ms with 8, 9.5 with 4,
//                            16 with 2; 45 ms with one wave per group in the first version of the scan).
static int pick_slices(int64_t point_tiles, int mesh_tiles) {
#ifdef PVAMD_FORCE_SLICES
    return PVAMD_FORCE_SLICES;
#endif
    if (point_tiles < (int64_t)kNumCU * 4) return 16;
    if (point_tiles < (int64_t)kNumCU * 16) return 8;  // 100k random points on the drill: 0.65 (16) / 0.56 (8) / 0.60 ms (4)
    return mesh_tiles > 128 ? 8 : 4;
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_points_aabb(const float* points, int64_t P, float* box_out, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (!box_out) return PVAMD_E_NULL;
    if (P > 0 && !points) return PVAMD_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    unsigned* codes = reinterpret_cast<unsigned*>(box_out);
    hipLaunchKernelGGL(aabb_init_kernel, dim3(1), dim3(64), 0, s, codes);
    if (P > 0) {
        const int64_t want = (P + 255) / 256;
        const unsigned blocks = want < 512 ? (unsigned)want : 512u;
        hipLaunchKernelGGL(aabb_reduce_kernel, dim3(blocks), dim3(256), 0, s, points, P, codes);
    }
    hipLaunchKernelGGL(aabb_decode_kernel, dim3(1), dim3(64), 0, s, codes);
    return (int)hipGetLastError();
}

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
                                int32_t* out_face, float* out_normal, void* scratch, void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!mesh || !out_dist || !out_grad) return PVAMD_E_NULL;
    if (mesh->F < 0) return PVAMD_E_SHAPE;
    if (!points || (mesh->F > 0 && (!mesh->rec || !mesh->tiles || !mesh->normal || !mesh->rec_of_face))) return PVAMD_E_NULL;
    if (scratch && !aligned_to(scratch, 8)) return PVAMD_E_ALIGN;
    const MeshArgs m = mesh_args(*mesh);
    const int64_t ptiles = (P + 63) / 64;
    if (ptiles > 0x7fffffff) return PVAMD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const QueryOut out{out_closest, out_dist, out_grad, out_face, out_normal};
    const int ntiles = (mesh->F + kTile - 1) / kTile;
    // few point groups, many tiles: spread each group's tiles over `parts` blocks (see mesh_query_first_kernel)
    // (A/B on the drill, 3k..60k random points: 16 waves x <=16 parts 0.175 / 0.222 / 0.391 / 0.539 ms; 8 waves x <=31 parts
    // 0.164 / 0.183 / 0.331 / 0.405; 4 waves x <=31 parts 0.227 / 0.214 / 0.281 / 0.394)
    const int split_waves = ptiles >= 256 ? 4 : (ptiles >= 32 ? 8 : 16);  // (1k points: 69 us with 16 waves, 87 with 8)
    int parts = (int)((int64_t)kNumCU * (split_waves == 4 ? 32 : 16) / ptiles);
    if (parts > 31) parts = 31;
    if (parts > ntiles / 4) parts = ntiles / 4;
    if (scratch && parts >= 2 && ptiles <= (int64_t)kNumCU * 8) {  // beyond ~130k points the single launch wins
        hipLaunchKernelGGL((mesh_query_first_kernel<16>), dim3((unsigned)ptiles), dim3(1024), 0, s, m, order, points, P, jitter_seed, index_base, scratch);
        if (split_waves == 4) hipLaunchKernelGGL((mesh_query_rest_kernel<4>), dim3((unsigned)ptiles, (unsigned)parts), dim3(256), 0, s, m, order, points, P, jitter_seed, index_base, scratch);
        else if (split_waves == 8) hipLaunchKernelGGL((mesh_query_rest_kernel<8>), dim3((unsigned)ptiles, (unsigned)parts), dim3(512), 0, s, m, order, points, P, jitter_seed, index_base, scratch);
        else hipLaunchKernelGGL((mesh_query_rest_kernel<16>), dim3((unsigned)ptiles, (unsigned)parts), dim3(1024), 0, s, m, order, points, P, jitter_seed, index_base, scratch);
        hipLaunchKernelGGL(mesh_query_finish_kernel, dim3((unsigned)ptiles), dim3(64), 0, s, m, order, points, P, scratch, out);
        return (int)hipGetLastError();
    }
    switch (pick_slices(ptiles, ntiles)) {
        case 8: hipLaunchKernelGGL((mesh_query_kernel<8>), dim3((unsigned)ptiles), dim3(512), 0, s, m, order, points, P, jitter_seed, index_base, out); break;
        case 4: hipLaunchKernelGGL((mesh_query_kernel<4>), dim3((unsigned)ptiles), dim3(256), 0, s, m, order, points, P, jitter_seed, index_base, out); break;
        default: hipLaunchKernelGGL((mesh_query_kernel<16>), dim3((unsigned)ptiles), dim3(1024), 0, s, m, order, points, P, jitter_seed, index_base, out); break;
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
        switch (pick_slices(ptiles * nb, (mesh->F + kTile - 1) / kTile)) {
            case 8: hipLaunchKernelGGL((chamfer_mesh_kernel<8>), dim3((unsigned)ptiles, nb), dim3(512), 0, s, m, order, W + 16 * (int64_t)b0, points, N, scale, out_sum + b0); break;
            case 4: hipLaunchKernelGGL((chamfer_mesh_kernel<4>), dim3((unsigned)ptiles, nb), dim3(256), 0, s, m, order, W + 16 * (int64_t)b0, points, N, scale, out_sum + b0); break;
            default: hipLaunchKernelGGL((chamfer_mesh_kernel<16>), dim3((unsigned)ptiles, nb), dim3(1024), 0, s, m, order, W + 16 * (int64_t)b0, points, N, scale, out_sum + b0); break;
        }
    }
    return (int)hipGetLastError();
}
