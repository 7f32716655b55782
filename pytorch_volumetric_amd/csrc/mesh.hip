// Point x triangle kernels (BASELINE configs C1, C5; also the voxel-cache build and LOOKUP_GT_SDF).
//   mesh_query:   closest surface point + ray-hit parity sign + gradient + face id   (reference sdf.py:122-172,
//                 where it is a device->host copy, two Embree BVH traversals on CPU threads, ~12 numpy passes)
//   chamfer_mesh: per-transform sum of (scale*d)^2 over the transformed points         (reference chamfer.py:79-94)
//
// fp32-VALU bound, not HBM bound.  Structure (details at "The scan" below):
//   * pvamd_mesh_prepare turns the soup into 96-byte records (bounding sphere, in-plane bounding rectangle, corners,
//     original face id), stored per tile of 256 as six planes of float4, + one bounding sphere per group of 16 and per
//     tile of 256 records.
//   * a block owns 64 neighbouring points (one per lane) and SLICES waves; every wave walks its own share of the tiles
//     without barriers.  The broad phase runs with lanes = tiles, then lanes = records, against a sphere around the
//     block's points: 64 bounds per vector instruction instead of one; only the few records that survive are tested
//     per point (very small P -> the tiles of a point group are spread over several blocks as well).
//   * conservative culling, tile -> group -> record sphere -> rectangle: a record is skipped when it provably cannot
//     lower a lane's best d^2 nor be hit by its ray.  The survivors are queued as (record, point) pairs and the exact
//     tests (the oracle's operation sequences) run densely over the queue.  Bit-identical to the plain double loop of
//     oracle/pvamd_oracle.c; with spatially sorted triangles and points this is a flat three-level BVH.
//   * ties in d^2 resolve to the lowest ORIGINAL face id (lexicographic min), independent of processing order.
#include <cstdlib>
#include "common.h"
#include "mesh_math.h"
#include "morton.h"
#include "order_small.h"
#include "wave_ops.h"

namespace pvamd {

constexpr int kRec = PVAMD_TRI_REC;      // floats per record
constexpr int kTile = PVAMD_TRI_TILE;    // records per tile: 256 * 96 B = 24 KB
constexpr int kGroup = PVAMD_TRI_GROUP;  // records per group: own bounding sphere
constexpr int kGroupsPerTile = kTile / kGroup;
static_assert(kTile == 256 && kGroup == 16 && kRec == 24, "the scan is written for 256-record tiles of six float4 planes");
// A record is six float4s; a tile stores them plane by plane ([plane][256] float4, 24 KB, whole tiles allocated), so that
// 64 lanes reading the same plane of 64 consecutive records move one contiguous KB:
//   plane 0  ctr, r       bounding sphere (ctr = centre of the in-plane bounding rectangle)
//   plane 1  u,   hu      unit vector along the longest edge, half extent of the triangle along it (about ctr)
//   plane 2  v,   hv      unit in-plane vector across it, half extent
//   plane 3  a,   face id (bits)
//   plane 4  b,   m0      m0: absolute slack of the rectangle test (plane fit + rounding of the frame)
//   plane 5  c,   0
// `tiles` buffer: [ntiles][4] tile spheres, then [ntiles][16][4] group spheres.
enum { kPlaneSphere = 0, kPlaneU = 1, kPlaneV = 2, kPlaneA = 3, kPlaneB = 4, kPlaneC = 5 };
constexpr int kTileFloats = kTile * kRec;
PVAMD_DEV const f32x4* tile_plane(const float* rec, int tile, int plane) {
    return reinterpret_cast<const f32x4*>(rec + (int64_t)tile * kTileFloats + plane * (kTile * 4));
}
PVAMD_DEV f32x4 record_plane(const float* rec, int j, int plane) { return tile_plane(rec, j / kTile, plane)[j % kTile]; }

struct MeshArgs {
    const float* normal;
    const float* rec;
    const float* tiles;
    const int* rec_of_face;
    int F;
    double ray_dir[3];
    unsigned long long* pairs;  // NULL, or two counters: exact closest-point tests | exact ray tests executed (pvamd_mesh_t::pair_counters)
};
// one global atomic per batch of exact tests, behind a wave-uniform test of a kernel argument: nothing when the caller did
// not ask (the benchmark asks in ONE untimed call, to price the work actually done against the fp32 peak: SURVEY.md 8(d))
#define PVAMD_COUNT_PAIRS(m, which, n)                                                                                  \
    do {                                                                                                                \
        if ((m).pairs) {                                                                                                \
            const unsigned long long count_ = (unsigned long long)(n); /* wave-uniform; evaluated by every active lane */ \
            if ((int)(threadIdx.x & 63) == __builtin_ctzll(__ballot(true))) atomicAdd((m).pairs + (which), count_);     \
        }                                                                                                               \
    } while (0)

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
    const int f = blockIdx.x * blockDim.x + threadIdx.x;  // one block per tile
    f32x4* o = reinterpret_cast<f32x4*>(rec + (int64_t)blockIdx.x * kTileFloats) + threadIdx.x;
    if (f >= F) {  // padding of the last tile: never a candidate (the kernels mask by F), but defined memory
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int pl = 0; pl < 6; ++pl) o[pl * kTile] = zero;
        return;
    }
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
    o[kPlaneSphere * kTile] = f32x4{cf[0], cf[1], cf[2], (float)(r * 1.00001) + abs_margin};
    o[kPlaneU * kTile] = f32x4{uf[0], uf[1], uf[2], (float)(hu * 1.00001) + abs_margin};
    o[kPlaneV * kTile] = f32x4{vf[0], vf[1], vf[2], (float)(hv * 1.00001) + abs_margin};
    o[kPlaneA * kTile] = f32x4{t[0], t[1], t[2], __int_as_float(face_id ? face_id[f] : f)};
    o[kPlaneB * kTile] = f32x4{t[3], t[4], t[5], (float)(1.01e6 * sl * sl) + abs_margin * abs_margin};
    o[kPlaneC * kTile] = f32x4{t[6], t[7], t[8], 0.f};
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
    const f32x4 o = tile_plane(rec, tile, kPlaneSphere)[live ? threadIdx.x : 0];
    const float cx = o.x, cy = o.y, cz = o.z, r = o.w;
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
PVAMD_DEV bool rect_may_improve(const LaneState& s, V3 w, float dist2, V3 fu, float hu, V3 fv, float hv, float m0) {
    const float u = dot(fu, w), v = dot(fv, w);
    const float h2 = fmaxf(fmaf(-v, v, fmaf(-u, u, dist2)), 0.f);
    const float du = fmaxf(fabsf(u) - hu, 0.f), dv = fmaxf(fabsf(v) - hv, 0.f);
    const float lb2 = fmaf(dv, dv, fmaf(du, du, h2));
    return !(lb2 > fmaf(8e-6f, dist2, s.reach2m + m0));
}

#ifdef PVAMD_MESH_STATS
__device__ unsigned long long g_stats[32];
__device__ long long g_nested;  // unused
#define STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
extern "C" int pvamd_debug_stats(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(g_stats));
    if (reset) { unsigned long long z[32] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)); }
    return 0;
}
#define TIC(t) const long long t = __builtin_readcyclecounter()
#define TOC(i, t) STAT(i, __builtin_readcyclecounter() - t)
#define TOC_NET(i, t, inner) STAT(i, __builtin_readcyclecounter() - t - (inner))
#else
#define STAT(i, v)
#define TIC(t)
#define TOC(i, t)
#define TOC_NET(i, t, inner)
#endif

// ---------------------------------------------------------------------------------------------------------------
// The scan.  One block = 64 points (one per lane; the caller passes a Morton order, so they are neighbours in space) x
// SLICES waves that all hold the same 64 points.  Wave w walks the tiles ti % SLICES == w on its own -- no barrier between
// the one that publishes the rays and the one before the results are read; the waves only meet in the per-point result
// slots (LDS atomics), from which each keeps pulling the others' finds, and every wave-uniform quantity sits in SGPRs.  (A
// point group near the medial axis needs most of the mesh; spreading its tiles over the waves bounds that tail, and the
// worst of them are handed over to a launch of their own: "point groups that are handed over" below.)
//   bound   the wave keeps a sphere (c, rho) around its live points and Q = max reach + rho.  Whatever a lane still
//           needs lies within Q of c: |p - ctr| >= |c - ctr| - rho, and the distance to a rectangle is 1-Lipschitz in p.
//           The same holds for the rays: all lanes shoot within `chord` of one direction, so a sphere missed by the
//           axis ray through c by more than r + rho + 2 chord (|w| + rho) is missed by every lane's ray.
//   tiles   lanes = TILES (64 tile spheres per pass): one evaluation of the sphere test at (c, Q) answers "does any
//           lane need this tile" for 64 tiles at once; the nearest tile is visited first, which brings every reach
//           down to about the true distance, then the flagged tiles of each pass nearest-first, re-flagged after every
//           visit and confirmed by one per-point look at the tile's sphere before the visit is paid for.
//   groups  lanes = the tile's 16 GROUP spheres at (c, Q); the few that pass are tested per point (sphere of group b
//           broadcast from lane b) -- what keeps the record passes to the groups some lane really needs.
//   records lanes = RECORDS (64 per pass, 4 passes per tile, skipped per pass through the group mask): sphere
//           test + rectangle test at (c, Q) + ray test about the axis, on coalesced 16-byte loads of the tile's planes.
//           Only the survivors (typically a few per cent) are looked at per point: their filter data goes through a
//           wave-private LDS slice and is read back wave-uniformly (LDS broadcast), one survivor at a time, for the
//           per-lane sphere / rectangle / ray tests, lanes = POINTS.
//   narrow  the per-point tests only QUEUE (record, point) pairs -- per wave, in LDS, one queue for closest-point pairs
//           and one for ray pairs, entries = global record index << 6 | owner lane, so a queue outlives the tile it was
//           filled from.  Whenever 64 are waiting the wave runs them densely: lane i takes pair i, gathers the three
//           corners from the record planes and its owner's point through ds_bpermute.  Results fold into per-point LDS
//           slots: a 64-bit atomicMin on (d2 bits << 32 | face id) IS the lexicographic "smallest d2, lowest original
//           face id" rule (d2 >= +0, so its bit pattern orders like the float; a NaN sorts above +inf and never wins),
//           hit counts by atomicAdd.  After a drain every lane tightens its reach and the wave its Q.
// A skipped record provably cannot lower a lane's best d^2 nor be hit by its ray, min and + are order-free, so the
// results are bit-identical to the plain double loop of oracle/pvamd_oracle.c.  A point with a NaN or infinite
// coordinate can neither find a finite d^2 nor hit anything (every product with it is inf or NaN): such lanes are
// "dead", take no part in the bounds and come out as (no face, 0 hits), as they do from the double loop.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kQueueCap = 128;       // entries per queue (u32: record << 6 | owner lane); < 64 + 64 in use
constexpr unsigned long long kBestInit = 0x7F80000000000000ull;  // (+inf, face 0): a finite candidate always wins

template <bool WITH_RAY>
struct WaveLocal {                      // private to one wave
    f32x4 sphere[64], fu[64], fv[64];   // filter planes of the current pass of 64 records
    float m0[64];
    unsigned qc[kQueueCap];
    unsigned qr[WITH_RAY ? kQueueCap : 1];
};
template <bool WITH_RAY>
struct GroupShared {                    // the per-point slots all waves of the point group fold into
    unsigned long long best[64];
    int hits[WITH_RAY ? 64 : 1];
    int handed;                         // the group was handed over (see scan_mesh)
    float dir[WITH_RAY ? 192 : 1];      // jittered ray direction (exact test), computed once by wave 0
    float dn[WITH_RAY ? 192 : 1];       // its unit vector (sphere tests)
};
template <int SLICES, bool WITH_RAY>
struct MeshShared {
    GroupShared<WITH_RAY> g;
    WaveLocal<WITH_RAY> w[SLICES];
};

PVAMD_DEV float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
PVAMD_DEV V3 xyz(f32x4 v) { return v3(v.x, v.y, v.z); }

struct WaveBound {   // wave-uniform
    LaneState q;     // p = c (centre of the live points' box), reach = Q = (max reach + rho), inflated
    float rho;       // max |p - c| over the live lanes, inflated
    V3 dn0;          // unit axis of the rays
    float k, rho_k;  // 2.001 * max |dn_lane - dn0|;  rho * (1 + k)
};

template <bool WITH_RAY>
struct Wave {
    LaneState s;  // per lane
    bool live;
    V3 dn, dir;   // unit / exact ray direction (WITH_RAY)
    WaveBound wb;
    int nc, nr;   // queue fills (wave-uniform)
};

template <bool WITH_RAY>
PVAMD_DEV void refresh_bound(Wave<WITH_RAY>& wv) {
    const float rmax = wave_max(wv.live ? wv.s.reach : 0.f);
    set_reach(wv.wb.q, uniform((rmax + wv.wb.rho) * 1.00001f));
}

// tighten this lane's reach from its result slot (other waves' finds included), then the wave's bound
template <bool WITH_RAY>
PVAMD_DEV void pull_reach(const GroupShared<WITH_RAY>& g, Wave<WITH_RAY>& wv) {
    const float d2 = __int_as_float((int)(unsigned)(g.best[threadIdx.x & 63] >> 32));
    const float reach = fast_sqrt(d2) * 1.00001f + 1.1e-19f;  // 1 ulp + the flushed denormals; inf while nothing was found
    if (reach < wv.s.reach) set_reach(wv.s, reach);
    refresh_bound(wv);
}

// per-wave setup once the lanes hold their points (and rays): dead lanes, (c, rho), the ray axis
template <bool WITH_RAY>
PVAMD_DEV bool wave_setup(const MeshArgs& m, Wave<WITH_RAY>& wv) {
    const V3 p = wv.s.p;
    wv.live = fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY;
    wv.nc = wv.nr = 0;
    set_reach(wv.s, INFINITY);
    if (__ballot(wv.live) == 0ull) return false;
    const float lx = wave_min(wv.live ? p.x : INFINITY), hx = wave_max(wv.live ? p.x : -INFINITY);
    const float ly = wave_min(wv.live ? p.y : INFINITY), hy = wave_max(wv.live ? p.y : -INFINITY);
    const float lz = wave_min(wv.live ? p.z : INFINITY), hz = wave_max(wv.live ? p.z : -INFINITY);
    wv.wb.q.p = v3(uniform(0.5f * lx + 0.5f * hx), uniform(0.5f * ly + 0.5f * hy), uniform(0.5f * lz + 0.5f * hz));
    const V3 d = v3(p.x - wv.wb.q.p.x, p.y - wv.wb.q.p.y, p.z - wv.wb.q.p.z);
    wv.wb.rho = uniform(wave_max(wv.live ? fast_sqrt(dot(d, d)) : 0.f) * 1.00001f + 1.1e-19f);
    wv.wb.dn0 = v3(0.f, 0.f, 0.f);
    wv.wb.k = wv.wb.rho_k = 0.f;
    if (WITH_RAY) {
        const V3 a = v3((float)m.ray_dir[0], (float)m.ray_dir[1], (float)m.ray_dir[2]);
        const float inv = 1.f / sqrt_rn(dot(a, a));
        wv.wb.dn0 = v3(uniform(a.x * inv), uniform(a.y * inv), uniform(a.z * inv));
        const V3 e = v3(wv.dn.x - wv.wb.dn0.x, wv.dn.y - wv.wb.dn0.y, wv.dn.z - wv.wb.dn0.z);
        const float chord = wave_max(wv.live ? fast_sqrt(dot(e, e)) : 0.f);  // a NaN (zero direction) is ignored: it hits nothing
        wv.wb.k = uniform(2.001f * chord + 2e-6f);
        wv.wb.rho_k = uniform(wv.wb.rho * (1.f + wv.wb.k) * 1.00001f);
    }
    set_reach(wv.wb.q, INFINITY);
    return true;
}

// may the axis ray, widened by everything the lanes' own rays can differ from it, cross the sphere (w = ctr - c, r)?
PVAMD_DEV bool axis_may_hit(const WaveBound& wb, V3 w, float dist2, float r) {
    const float dist = fast_sqrt(dist2) * 1.00001f + 1.1e-19f;
    return sphere_may_hit(dist2, dot(w, wb.dn0), fmaf(wb.k, dist, r + wb.rho_k));
}

// enqueue the lanes of `mask` for record j (wave-uniform j): the k-th set lane writes slot n + k
PVAMD_DEV void enqueue(unsigned* q, int& n, unsigned long long mask, bool mine, int j) {
    const int lane = threadIdx.x & 63;
    if (mine) q[n + __popcll(mask & ((1ull << lane) - 1ull))] = ((unsigned)j << 6) | (unsigned)lane;
    n += __popcll(mask);
}

// Run fn(count, entries) densely over the queue: always the first (up to) 64 entries; the remainder (n - 64 < 64 by
// construction) too when `everything`, else it moves to the front and waits for more company.
template <class Fn>
PVAMD_DEV void drain(unsigned* q, int& n, bool everything, Fn&& fn) {
    const int lane = threadIdx.x & 63;
    PVAMD_WAVE_SYNC();  // entries were written by other lanes
    STAT(6, 1);
    fn(n < 64 ? n : 64, q);
    if (n > 64) {
        const int rem = n - 64;
        if (everything) {
            STAT(6, 1);
            fn(rem, q + 64);
            n = 0;
        } else {
            const unsigned v = lane < rem ? q[64 + lane] : 0u;
            if (lane < rem) q[lane] = v;
            n = rem;
        }
    } else {
        n = 0;
    }
    PVAMD_WAVE_SYNC();
}

template <bool WITH_RAY>
PVAMD_DEV void drain_closest(const MeshArgs& m, GroupShared<WITH_RAY>& g, WaveLocal<WITH_RAY>& wl, Wave<WITH_RAY>& wv, bool everything) {
    if (wv.nc == 0) return;
    const int lane = threadIdx.x & 63;
    TIC(t_drain);
    drain(wl.qc, wv.nc, everything, [&](int count, const unsigned* q) {
        PVAMD_COUNT_PAIRS(m, 0, count);
        const unsigned e = lane < count ? q[lane] : 0u;
        const int owner = e & 63;
        const V3 p = v3(__shfl(wv.s.p.x, owner, 64), __shfl(wv.s.p.y, owner, 64), __shfl(wv.s.p.z, owner, 64));
        if (lane < count) {
            const int j = (int)(e >> 6);
            const f32x4 A = record_plane(m.rec, j, kPlaneA), B = record_plane(m.rec, j, kPlaneB), C = record_plane(m.rec, j, kPlaneC);
            const V3 qp = sub(closest_point_triangle(p, xyz(A), xyz(B), xyz(C)), p);
            const float d2 = dot(qp, qp);
            atomicMin(&g.best[owner], ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(A.w));
        }
    });
    pull_reach(g, wv);
    TOC(16, t_drain);
}

template <bool WITH_RAY>
PVAMD_DEV void drain_rays(const MeshArgs& m, GroupShared<WITH_RAY>& g, WaveLocal<WITH_RAY>& wl, Wave<WITH_RAY>& wv, bool everything) {
    if (wv.nr == 0) return;
    const int lane = threadIdx.x & 63;
    drain(wl.qr, wv.nr, everything, [&](int count, const unsigned* q) {
        PVAMD_COUNT_PAIRS(m, 1, count);
        const unsigned e = lane < count ? q[lane] : 0u;
        const int owner = e & 63;
        const V3 p = v3(__shfl(wv.s.p.x, owner, 64), __shfl(wv.s.p.y, owner, 64), __shfl(wv.s.p.z, owner, 64));
        const V3 d = v3(__shfl(wv.dir.x, owner, 64), __shfl(wv.dir.y, owner, 64), __shfl(wv.dir.z, owner, 64));
        if (lane < count) {
            const int j = (int)(e >> 6);
            const f32x4 A = record_plane(m.rec, j, kPlaneA), B = record_plane(m.rec, j, kPlaneB), C = record_plane(m.rec, j, kPlaneC);
            if (ray_hits_triangle(p, d, xyz(A), xyz(B), xyz(C))) atomicAdd(&g.hits[owner], 1);
        }
    });
}

// One tile: group spheres -> record passes at (c, Q) -> survivors per point -> queues.
template <bool WITH_RAY>
PVAMD_DEV void visit_tile(const MeshArgs& m, GroupShared<WITH_RAY>& g, WaveLocal<WITH_RAY>& wl, Wave<WITH_RAY>& wv, int ti,
                          int pass_lo = 0, int pass_hi = kTile / 64) {
    const int lane = threadIdx.x & 63;
    const int ntiles = (m.F + kTile - 1) / kTile;
    const int n = min(kTile, m.F - ti * kTile);
    STAT(1, 1);
    TIC(t_visit);
    unsigned gm = 0u;
#ifdef PVAMD_MESH_STATS
    unsigned mine_groups = 0u;
#endif
    {
        // lanes = groups: the 16 group spheres at (c, Q) ...
        const bool gv = lane < kGroupsPerTile && lane * kGroup < n;
        const f32x4 gs = reinterpret_cast<const f32x4*>(m.tiles)[ntiles + ti * kGroupsPerTile + (gv ? lane : 0)];
        const V3 wc = v3(gs.x - wv.wb.q.p.x, gs.y - wv.wb.q.p.y, gs.z - wv.wb.q.p.z);
        const float dc2 = dot(wc, wc);
        bool coarse = sphere_may_improve(wv.wb.q, dc2, gs.w);
        if (WITH_RAY) coarse = coarse || axis_may_hit(wv.wb, wc, dc2, gs.w);
        unsigned todo = (unsigned)__ballot(coarse && gv);
        todo &= ((1u << (pass_hi * (64 / kGroup))) - 1u) & ~((1u << (pass_lo * (64 / kGroup))) - 1u);  // this wave's passes
        // ... then, for the ones that pass, lanes = points: does any lane need the group?  (sphere of group b from lane b)
        while (todo != 0u) {
            const int b = __builtin_ctz(todo);
            todo &= todo - 1u;
            const V3 ctr = v3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(gs.x), b)),
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gs.y), b)),
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gs.z), b)));
            const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gs.w), b));
            const V3 w = v3(ctr.x - wv.s.p.x, ctr.y - wv.s.p.y, ctr.z - wv.s.p.z);
            const float dist2 = dot(w, w);
            bool need = sphere_may_improve(wv.s, dist2, r);
            if (WITH_RAY) need = need || sphere_may_hit(dist2, dot(w, wv.dn), r);
            STAT(9, 1);
#ifdef PVAMD_MESH_STATS
            if (need && wv.live) mine_groups |= 1u << b;
#endif
            if (__ballot(need && wv.live) != 0ull) gm |= 1u << b;  // (the ballot IS the compare mask: __any materialises 0 / 1 and compares again)
        }
    }
    const f32x4* P0 = tile_plane(m.rec, ti, kPlaneSphere);
    const f32x4* P1 = tile_plane(m.rec, ti, kPlaneU);
    const f32x4* P2 = tile_plane(m.rec, ti, kPlaneV);
    const f32x4* P4 = tile_plane(m.rec, ti, kPlaneB);
    for (int pass = pass_lo; pass < pass_hi; ++pass) {
        if (((gm >> (pass * (64 / kGroup))) & ((1u << (64 / kGroup)) - 1u)) == 0u) continue;
        STAT(2, 1);
#ifdef PVAMD_MESH_STATS
        {
            const int st_points = __popcll(__ballot(((mine_groups >> (pass * (64 / kGroup))) & 0xFu) != 0u));
            STAT(22, st_points);
        }
#endif
        const int idx = pass * 64 + lane;
        const f32x4 a0 = P0[idx], a1 = P1[idx], a2 = P2[idx];
        const float am0 = P4[idx].w;
        bool keep_c, keep_r = false;
        {
            const bool valid = idx < n && ((gm >> (idx / kGroup)) & 1u);
            const V3 w = v3(a0.x - wv.wb.q.p.x, a0.y - wv.wb.q.p.y, a0.z - wv.wb.q.p.z);
            const float dist2 = dot(w, w);
            keep_c = valid && sphere_may_improve(wv.wb.q, dist2, a0.w) &&
                     rect_may_improve(wv.wb.q, w, dist2, xyz(a1), a1.w, xyz(a2), a2.w, am0);
            if (WITH_RAY) keep_r = valid && axis_may_hit(wv.wb, w, dist2, a0.w);
        }
        const unsigned long long mc = __ballot(keep_c), mr = WITH_RAY ? __ballot(keep_r) : 0ull;
        unsigned long long todo = mc | mr;
        if (todo == 0ull) continue;
        PVAMD_WAVE_SYNC();  // the previous pass's broadcast reads are done
        wl.sphere[lane] = a0;
        wl.fu[lane] = a1;
        wl.fv[lane] = a2;
        wl.m0[lane] = am0;
        PVAMD_WAVE_SYNC();
        const int j0 = ti * kTile + pass * 64;
        const unsigned long long live_mask = __ballot(wv.live);
        TIC(t_surv);
        while (todo != 0ull) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            STAT(3, 1);
            const f32x4 o = wl.sphere[b];  // wave-uniform address: LDS broadcast
            const V3 w = v3(o.x - wv.s.p.x, o.y - wv.s.p.y, o.z - wv.s.p.z);
            const float dist2 = dot(w, w);
            // Lane predicates stay LANE MASKS (the compare's own SGPR pair) from the test to the queue: a bool that crosses a
            // branch is materialised as 0 / 1 in a VGPR and compared again at every use (two vector instructions per survivor
            // in the sphere stage and two more in the rectangle stage: 11 % of this kernel's vector instructions on C5), and
            // `a && f(...)` wraps f in an exec-mask save / restore.  The rectangle test runs for every lane; the masks are ANDed.
            if (!WITH_RAY || ((mc >> b) & 1ull)) {  // (without rays every survivor is a closest-point candidate: todo == mc)
                const unsigned long long near_c = live_mask & __ballot(sphere_may_improve(wv.s, dist2, o.w));
                if (near_c != 0ull) {
                    STAT(7, 1);
                    const f32x4 u = wl.fu[b], v = wl.fv[b];
                    const unsigned long long mk = near_c & __ballot(rect_may_improve(wv.s, w, dist2, xyz(u), u.w, xyz(v), v.w, wl.m0[b]));
                    if (mk) {
                        STAT(4, __popcll(mk)); STAT(8, 1);
                        enqueue(wl.qc, wv.nc, mk, __builtin_amdgcn_inverse_ballot_w64(mk), j0 + b);
                        if (wv.nc >= 64) drain_closest(m, g, wl, wv, false);
                    }
                }
            }
            if (WITH_RAY && ((mr >> b) & 1ull)) {
                const unsigned long long mk = live_mask & __ballot(sphere_may_hit(dist2, dot(w, wv.dn), o.w));
                if (mk) {
                    STAT(5, __popcll(mk)); STAT(8, 1);
                    enqueue(wl.qr, wv.nr, mk, __builtin_amdgcn_inverse_ballot_w64(mk), j0 + b);
                    if (wv.nr >= 64) drain_rays(m, g, wl, wv, false);
                }
            }
        }
        TOC(18, t_surv);
    }
    TOC(17, t_visit);
}

// Upper bound on every live lane's distance to the mesh from the tile spheres (each contains whole triangles):
// min over ALL tiles of |c - ctr| + r, plus rho.  Returns the nearest of the wave's OWN tiles (ti % nparts == part) -- the
// one it visits first -- or -1 when it owns none.
template <bool WITH_RAY>
PVAMD_DEV int scan_seed(const MeshArgs& m, Wave<WITH_RAY>& wv, int part, int nparts) {
    const int lane = threadIdx.x & 63;
    const int ntiles = (m.F + kTile - 1) / kTile;
    const f32x4* tiles4 = reinterpret_cast<const f32x4*>(m.tiles);
    float bound = INFINITY, own = INFINITY;
    int nearest = -1;
    for (int base = 0; base < ntiles; base += 64) {
        const int ti = base + lane;
        if (ti < ntiles) {
            const f32x4 ts = tiles4[ti];
            const V3 w = v3(ts.x - wv.wb.q.p.x, ts.y - wv.wb.q.p.y, ts.z - wv.wb.q.p.z);
            const float b = fast_sqrt(dot(w, w)) * 1.00001f + (ts.w + 1.1e-19f);  // slack: 1 ulp + the flushed denormals
            bound = fminf(bound, b);  // a NaN never lowers it
            if ((nparts == 1 || (ti % nparts) == part) && (b < own || nearest < 0)) {
                own = b < own ? b : own;
                nearest = ti;
            }
        }
    }
    const float lowest = wave_min(bound);
    set_reach(wv.s, (lowest + wv.wb.rho) * 1.00001f);
    refresh_bound(wv);
    const float mine = wave_min(own);
    unsigned long long at = __ballot(nearest >= 0 && own == mine);
    if (at == 0ull) at = __ballot(nearest >= 0);  // only NaN bounds: any own tile
    return at ? __builtin_amdgcn_readlane(nearest, __builtin_ctzll(at)) : -1;
}

// Before the first visit: tighten every lane's reach by a greedy descent through tile `ti` -- the group, then the record,
// with the smallest |p - ctr| + r (each sphere contains whole triangles, so each is an upper bound of the lane's distance
// to the mesh).  The first visit then queues pairs against a bound a few millimetres wide instead of a tile radius.
template <bool WITH_RAY>
PVAMD_DEV void greedy_reach(const MeshArgs& m, GroupShared<WITH_RAY>& g, Wave<WITH_RAY>& wv, int ti) {
    const int ntiles = (m.F + kTile - 1) / kTile;
    const f32x4* spheres = reinterpret_cast<const f32x4*>(m.tiles);
    const V3 p = wv.s.p;
    auto reach_of = [&](f32x4 sp) {
        const V3 w = v3(sp.x - p.x, sp.y - p.y, sp.z - p.z);
        return fast_sqrt(dot(w, w)) * 1.00001f + (sp.w + 1.1e-19f);
    };
    float bound = INFINITY;
    int gi = 0;
    const int ngroups = (min(kTile, m.F - ti * kTile) + kGroup - 1) / kGroup;
    for (int k = 0; k < ngroups; ++k) {  // wave-uniform: scalar loads
        const float b = reach_of(spheres[ntiles + ti * kGroupsPerTile + k]);
        if (b < bound) { bound = b; gi = k; }
    }
    const int j0 = ti * kTile + gi * kGroup;  // per lane
    float nearest = INFINITY;
    int jn = -1;
    for (int k = 0; k < kGroup; ++k) {
        if (j0 + k < m.F) {
            const float b = reach_of(record_plane(m.rec, j0 + k, kPlaneSphere));
            if (b < nearest) { nearest = b; jn = j0 + k; }
        }
    }
#ifndef PVAMD_MESH_NO_GREEDY_EXACT
    // ... and the exact distance to that record's triangle, all 64 lanes at once (what a drain does for 64 queued pairs):
    // the bound drops from "the far side of the nearest record's sphere" to a real distance before anything is queued
    // against it, and the pair is a candidate like any other (same operations as in the drain: same bits).
    PVAMD_COUNT_PAIRS(m, 0, __popcll(__ballot(jn >= 0 && wv.live)));
    if (jn >= 0 && wv.live) {
        const f32x4 A = record_plane(m.rec, jn, kPlaneA), B = record_plane(m.rec, jn, kPlaneB), C = record_plane(m.rec, jn, kPlaneC);
        const V3 qp = sub(closest_point_triangle(p, xyz(A), xyz(B), xyz(C)), p);
        const float d2 = dot(qp, qp);
        atomicMin(&g.best[threadIdx.x & 63], ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(A.w));
        STAT(4, 1);
    }
#endif
    bound = fminf(bound, nearest) * 1.00001f;
    if (bound < wv.s.reach) set_reach(wv.s, bound);  // a NaN point: never
    pull_reach(g, wv);
}

// the tiles ti with ti % nparts == part, `skip` excepted (it was visited before), flagged 64 at a time
template <bool WITH_RAY>
PVAMD_DEV void scan_tiles(const MeshArgs& m, GroupShared<WITH_RAY>& g, WaveLocal<WITH_RAY>& wl, Wave<WITH_RAY>& wv, int skip,
                          int part, int nparts, int pass_lo = 0, int pass_hi = kTile / 64) {
    const int lane = threadIdx.x & 63;
    const int ntiles = (m.F + kTile - 1) / kTile;
    const f32x4* tiles4 = reinterpret_cast<const f32x4*>(m.tiles);
    for (int base = 0; base < ntiles; base += 64) {
        const int ti = base + lane;
        const bool mine = ti < ntiles && ti != skip && (nparts == 1 || (ti % nparts) == part);
        const f32x4 ts = tiles4[mine ? ti : 0];
        const V3 w = v3(ts.x - wv.wb.q.p.x, ts.y - wv.wb.q.p.y, ts.z - wv.wb.q.p.z);
        const float dist2 = dot(w, w);
        const bool ray = WITH_RAY && mine && axis_may_hit(wv.wb, w, dist2, ts.w);
        STAT(0, 1);
        // nearest flagged tile of the pass first (the sooner the reaches are final, the fewer pairs get queued), and a
        // per-point look at its sphere before paying for the visit
        const float lower = mine ? fast_sqrt(dist2) - ts.w : INFINITY;  // ordering only
        unsigned long long done = 0ull;
        for (;;) {
            pull_reach(g, wv);  // what the other waves found in the meantime
            const bool need = mine && (ray || sphere_may_improve(wv.wb.q, dist2, ts.w));  // Q only ever shrinks
            const unsigned long long todo = __ballot(need) & ~done;
            if (todo == 0ull) break;
            const float key = ((todo >> lane) & 1ull) ? lower : INFINITY;
            const unsigned long long at = __ballot(key == wave_min(key)) & todo;
            const int t = __builtin_ctzll(at ? at : todo);
            done |= 1ull << t;
            const V3 ctr = v3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ts.x), t)),
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ts.y), t)),
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ts.z), t)));
            const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ts.w), t));
            const V3 wl_ = v3(ctr.x - wv.s.p.x, ctr.y - wv.s.p.y, ctr.z - wv.s.p.z);
            const float d2 = dot(wl_, wl_);
            bool mine_too = sphere_may_improve(wv.s, d2, r);
            if (WITH_RAY) mine_too = mine_too || sphere_may_hit(d2, dot(wl_, wv.dn), r);
            if (__ballot(mine_too && wv.live) == 0ull) continue;
            visit_tile<WITH_RAY>(m, g, wl, wv, base + t, pass_lo, pass_hi);
        }
    }
}

template <bool WITH_RAY>
PVAMD_DEV void scan_finish(const MeshArgs& m, GroupShared<WITH_RAY>& g, WaveLocal<WITH_RAY>& wl, Wave<WITH_RAY>& wv) {
    drain_closest(m, g, wl, wv, true);
    if (WITH_RAY) drain_rays(m, g, wl, wv, true);
}

// Block prologue: wave 0 publishes the result slots (and the rays); on return every wave holds its rays and bounds.
// False when no lane has a finite point (uniform over the block: every wave holds the same points).
template <bool WITH_RAY>
PVAMD_DEV bool scan_begin(const MeshArgs& m, GroupShared<WITH_RAY>& g, Wave<WITH_RAY>& wv, int wave, uint64_t seed,
                          int64_t jitter_index, const unsigned long long* start, const float* __restrict__ drawn = nullptr,
                          const float* __restrict__ bound = nullptr) {
    const int lane = threadIdx.x & 63;
    if (wave == 0) {
        g.best[lane] = start ? *start : kBestInit;
        if (WITH_RAY) {
            g.hits[lane] = 0;
            // the jitter is ~1200 instructions of hashing on the block's critical path: a listed group finds it drawn already
            const V3 dir = drawn ? v3(drawn[3 * lane], drawn[3 * lane + 1], drawn[3 * lane + 2])
                                 : jitter_dir(m.ray_dir, seed, jitter_index);
            const float inv_len = 1.f / sqrt_rn(dot(dir, dir));
            g.dir[3 * lane] = dir.x; g.dir[3 * lane + 1] = dir.y; g.dir[3 * lane + 2] = dir.z;
            g.dn[3 * lane] = dir.x * inv_len; g.dn[3 * lane + 1] = dir.y * inv_len; g.dn[3 * lane + 2] = dir.z * inv_len;
        }
    }
    __syncthreads();
    wv.dir = wv.dn = v3(0.f, 0.f, 0.f);
    if (WITH_RAY) {
        wv.dir = v3(g.dir[3 * lane], g.dir[3 * lane + 1], g.dir[3 * lane + 2]);
        wv.dn = v3(g.dn[3 * lane], g.dn[3 * lane + 1], g.dn[3 * lane + 2]);
    }
    return bound ? load_bound(bound, wv) : wave_setup(m, wv);
}

// ---- point groups that are handed over -----------------------------------------------------------------------
// The work per point group is heavy-tailed: a group near the medial axis is about equidistant to much of the surface and
// needs most of the mesh (C5: the group around the sphere's centre runs 5.8 ms on its own 8 waves while the other
// 32,767 groups are done in 4.4 ms).  Such a group -- after its nearest tiles, with the reaches about final, it still
// flags most of the tiles -- is HANDED OVER: it appends itself to a list in the caller's scratch and stops; a
// second launch spreads the tiles of every listed group over kHeavyParts blocks of four waves (one per 64-record pass),
// folding into the group's scratch slots with global atomicMin / atomicAdd, and a third writes the listed groups'
// outputs.  The few-points path (below) lists EVERY group up front, in a launch of its own, and needs no third launch.
// scratch: int count | int entries[cap][2] (point group, transform) | u64 best[cap][64] | int hits[cap][64] |
//          float dir[cap][64][3] (the jittered ray of each point, drawn once) | float reach[cap][64] (an upper bound
//          of each point's distance to the mesh to start from, +inf when none was worked out) | float bound[cap][16]
//          (the group's wave bound: every block of the parts launch would otherwise redo its eight wave reductions) |
//          float pts[cap][64][3] (the group's points)
struct HandOver {
    int* count;
    int* entries;
    unsigned long long* best;
    int* hits;
    float* dir;
    float* reach;
    float* pts;    // [cap][64][3]: the group's points (for the chamfer calls: transformed), so that the blocks of the parts launch
                   // read them with one coalesced load instead of the order -> point chain
    // by POINT INDEX (queries of up to kSmallPoints points that bring no order: pvamd_mesh_query_unordered), filled by the
    // launch that also sorts, gathered into the slots by the next one
    unsigned long long* p_best;  // [kSmallPoints]
    float* p_dir;                // [kSmallPoints][3]
    float* p_reach;              // [kSmallPoints]
    float* bound;  // [cap][16]: the group's wave bound (c, rho, ray axis, k, rho_k, any lane live), worked out once when it is listed
    int cap;  // 0: nothing is handed over
};
constexpr int kBoundFloats = 16;
constexpr int kSmallPoints = PVAMD_MESH_SMALL_POINTS;
constexpr int kHandOverHeader = 64;  // bytes
static __host__ __device__ inline HandOver hand_over(void* scratch, int cap) {
    HandOver h;
    char* base = reinterpret_cast<char*>(scratch);
    h.count = reinterpret_cast<int*>(base);
    h.entries = reinterpret_cast<int*>(base + kHandOverHeader);
    h.best = reinterpret_cast<unsigned long long*>(base + kHandOverHeader + (size_t)cap * 8);
    h.hits = reinterpret_cast<int*>(base + kHandOverHeader + (size_t)cap * 8 + (size_t)cap * 64 * 8);
    h.dir = reinterpret_cast<float*>(base + kHandOverHeader + (size_t)cap * 8 + (size_t)cap * 64 * 12);
    h.reach = reinterpret_cast<float*>(base + kHandOverHeader + (size_t)cap * 8 + (size_t)cap * 64 * 24);
    h.bound = reinterpret_cast<float*>(base + kHandOverHeader + (size_t)cap * 8 + (size_t)cap * 64 * 28);
    h.pts = reinterpret_cast<float*>(base + kHandOverHeader + (size_t)cap * 8 + (size_t)cap * 64 * 28 + (size_t)cap * 64);
    char* tail = base + kHandOverHeader + (size_t)cap * (8 + 64 * 40 + 64);
    h.p_best = reinterpret_cast<unsigned long long*>(tail);
    h.p_dir = reinterpret_cast<float*>(tail + (size_t)kSmallPoints * 8);
    h.p_reach = reinterpret_cast<float*>(tail + (size_t)kSmallPoints * 20);
    h.cap = scratch ? cap : 0;
    return h;
}
#ifndef PVAMD_MESH_HEAVY_32NDS
#define PVAMD_MESH_HEAVY_32NDS 6
#endif
#ifndef PVAMD_MESH_HEAVY_PARTS
#define PVAMD_MESH_HEAVY_PARTS 32
#endif
// heavy = still flags this many 32nds of the tiles (of at least kHeavyMinTiles) once its reaches are about final.  C5
// (389 tiles, 32,768 groups), round 4 (triangles in patches, points along the Hilbert curve, exact greedy bound), whole
// call / groups listed with 2 waves per group in the main launch: 4/32 3.24 ms / 1782, 5/32 3.20 / 1000, 6/32 3.17 / 745,
// 7/32 3.18 / 569, 8/32 3.47 / 470, 10/32 3.47 / 301; with 4 waves per group 8/32 3.55, 12/32 3.77, 16/32 3.96 (64 instead
// of 32 parts: -0.05 with 4 waves, +0.1 with 2); nothing handed over: 5.15 (8 waves).  (Round 3, Z-order: 12/32 and 4
// waves, 3.71.)
constexpr int kHeavy32nds = PVAMD_MESH_HEAVY_32NDS;
#ifndef PVAMD_MESH_HEAVY_MIN_TILES
#define PVAMD_MESH_HEAVY_MIN_TILES 128
#endif
constexpr int kHeavyMinTiles = PVAMD_MESH_HEAVY_MIN_TILES;
constexpr int kHeavyParts = PVAMD_MESH_HEAVY_PARTS;

PVAMD_DEV void store_bound(float* __restrict__ b, const WaveBound& wb, bool any_live) {
    if ((threadIdx.x & 63) != 0) return;
    b[0] = wb.q.p.x; b[1] = wb.q.p.y; b[2] = wb.q.p.z; b[3] = wb.rho;
    b[4] = wb.dn0.x; b[5] = wb.dn0.y; b[6] = wb.dn0.z; b[7] = wb.k;
    b[8] = wb.rho_k; b[9] = any_live ? 1.f : 0.f;
    b[10] = wb.q.reach;  // Q when the group was listed (+inf: unknown)
}
// Does the block of part `part` of `nparts` have anything to do for the listed group whose bound is `b`?  The wave-level
// test of scan_tiles at the group's (c, Q) on the block's own tiles, before anything else is loaded: about half the blocks
// of a few-points query own no tile that any lane can need, and go straight to their tick.
template <bool WITH_RAY>
PVAMD_DEV bool part_has_work(const MeshArgs& m, const float* __restrict__ b, int part, int nparts) {
    if (uniform(b[9]) == 0.f) return false;
    WaveBound wb;
    wb.q.p = v3(uniform(b[0]), uniform(b[1]), uniform(b[2]));
    wb.rho = uniform(b[3]);
    wb.dn0 = v3(uniform(b[4]), uniform(b[5]), uniform(b[6]));
    wb.k = uniform(b[7]);
    wb.rho_k = uniform(b[8]);
    set_reach(wb.q, uniform(b[10]));
    const int lane = threadIdx.x & 63;
    const int ntiles = (m.F + kTile - 1) / kTile;
    const f32x4* tiles4 = reinterpret_cast<const f32x4*>(m.tiles);
    for (int base = part; base < ntiles; base += 64 * nparts) {
        const int ti = base + lane * nparts;
        const f32x4 ts = tiles4[ti < ntiles ? ti : 0];
        const V3 w = v3(ts.x - wb.q.p.x, ts.y - wb.q.p.y, ts.z - wb.q.p.z);
        const float dist2 = dot(w, w);
        bool need = sphere_may_improve(wb.q, dist2, ts.w);
        if (WITH_RAY) need = need || axis_may_hit(wb, w, dist2, ts.w);
        if (__ballot(need && ti < ntiles) != 0ull) return true;
    }
    return false;
}
// the per-wave setup of a listed group: the lanes' own state + the stored bound.  False: no lane has a finite point.
template <bool WITH_RAY>
PVAMD_DEV bool load_bound(const float* __restrict__ b, Wave<WITH_RAY>& wv) {
    const V3 p = wv.s.p;
    wv.live = fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY;
    wv.nc = wv.nr = 0;
    set_reach(wv.s, INFINITY);
    wv.wb.q.p = v3(uniform(b[0]), uniform(b[1]), uniform(b[2]));
    wv.wb.rho = uniform(b[3]);
    wv.wb.dn0 = v3(uniform(b[4]), uniform(b[5]), uniform(b[6]));
    wv.wb.k = uniform(b[7]);
    wv.wb.rho_k = uniform(b[8]);
    set_reach(wv.wb.q, INFINITY);
    return uniform(b[9]) != 0.f;
}

// tiles some lane may still need, counted with lanes = tiles at (c, Q)
template <bool WITH_RAY>
PVAMD_DEV int flagged_tiles(const MeshArgs& m, const Wave<WITH_RAY>& wv) {
    const int lane = threadIdx.x & 63;
    const int ntiles = (m.F + kTile - 1) / kTile;
    const f32x4* tiles4 = reinterpret_cast<const f32x4*>(m.tiles);
    int n = 0;
    for (int base = 0; base < ntiles; base += 64) {
        const int ti = base + lane;
        const f32x4 ts = tiles4[ti < ntiles ? ti : 0];
        const V3 w = v3(ts.x - wv.wb.q.p.x, ts.y - wv.wb.q.p.y, ts.z - wv.wb.q.p.z);
        const float dist2 = dot(w, w);
        bool need = sphere_may_improve(wv.wb.q, dist2, ts.w);
        if (WITH_RAY) need = need || axis_may_hit(wv.wb, w, dist2, ts.w);
        n += __popcll(__ballot(need && ti < ntiles));
    }
    return n;
}

// The whole scan of one block.  On exit (after a barrier): g.best[lane] / g.hits[lane] = result for point `lane` --
// unless the group was handed over (returns true): then the block has nothing to report.
template <int SLICES, bool WITH_RAY>
PVAMD_DEV bool scan_mesh(const MeshArgs& m, MeshShared<SLICES, WITH_RAY>& sh, Wave<WITH_RAY>& wv, int wave, uint64_t seed,
                         int64_t jitter_index, const HandOver& ho, int group, int transform) {
    const int lane = threadIdx.x & 63;
    bool handed = false;
    TIC(t_all);
    if (scan_begin(m, sh.g, wv, wave, seed, jitter_index, nullptr) && m.F > 0) {
        TIC(t_seed);
        const int first = scan_seed(m, wv, wave, SLICES);
        if (first >= 0) {
#ifndef PVAMD_MESH_NO_GREEDY
            greedy_reach(m, sh.g, wv, first);
#endif
            TOC(20, t_seed);
            visit_tile<WITH_RAY>(m, sh.g, sh.w[wave], wv, first);
            drain_closest(m, sh.g, sh.w[wave], wv, true);  // publish what the nearest tile gave before looking further
        }
        if (ho.cap > 0 && (m.F + kTile - 1) / kTile >= kHeavyMinTiles) {  // uniform over the block
            __syncthreads();  // every wave's nearest tile is in the slots
            if (wave == 0) {
                pull_reach(sh.g, wv);
                int slot = -1;
                if (flagged_tiles(m, wv) * 32 >= ((m.F + kTile - 1) / kTile) * kHeavy32nds) {
                    if (lane == 0) slot = atomicAdd(ho.count, 1);
                    slot = __builtin_amdgcn_readfirstlane(slot);
                    if (slot >= ho.cap) slot = -1;  // list full: this block does the work itself
                }
                if (slot >= 0) {
                    if (lane == 0) {
                        ho.entries[2 * slot] = group;
                        ho.entries[2 * slot + 1] = transform;
                    }
                    ho.best[(int64_t)slot * 64 + lane] = sh.g.best[lane];  // a bound to start from; the hits are counted afresh
                    ho.reach[(int64_t)slot * 64 + lane] = INFINITY;
                    store_bound(ho.bound + (int64_t)slot * kBoundFloats, wv.wb, true);
                    {
                        float* q = ho.pts + ((int64_t)slot * 64 + lane) * 3;
                        q[0] = wv.s.p.x; q[1] = wv.s.p.y; q[2] = wv.s.p.z;
                    }
                    if (WITH_RAY) {
                        ho.hits[(int64_t)slot * 64 + lane] = 0;
                        for (int d = 0; d < 3; ++d) ho.dir[((int64_t)slot * 64 + lane) * 3 + d] = sh.g.dir[3 * lane + d];
                    }
                }
                if (lane == 0) sh.g.handed = slot >= 0 ? 1 : 0;
            }
            __syncthreads();
            handed = sh.g.handed != 0;
        }
        if (!handed && first >= 0) {
            scan_tiles<WITH_RAY>(m, sh.g, sh.w[wave], wv, first, wave, SLICES);
            scan_finish(m, sh.g, sh.w[wave], wv);
        }
    }
    TOC(19, t_all);
    __syncthreads();
    return handed;
}

// the closest point on the winning face, recomputed from its corners (same operations as during the scan)
PVAMD_DEV V3 closest_on_record(const MeshArgs& m, int j, V3 p) {
    return closest_point_triangle(p, xyz(record_plane(m.rec, j, kPlaneA)), xyz(record_plane(m.rec, j, kPlaneB)),
                                  xyz(record_plane(m.rec, j, kPlaneC)));
}

struct QueryOut {
    float* closest;
    float* dist;
    float* grad;
    int* face;
    float* normal;
    float* packed;  // NULL, or [P][4] (dist, gx, gy, gz) records instead of dist / grad (pvamd_cache_build: the voxel cache itself)
};

// the index (caller order) of the point at position `k` of the processing order
PVAMD_DEV int64_t point_index(const int* __restrict__ order, int64_t k, int64_t P) {
    const int64_t kk = k < P ? k : (P - 1);
    return order ? (int64_t)order[kk] : kk;  // spatially sorted processing; outputs stay in caller order
}

// sdf.py:139-171 from the winning (d2, face) and the hit count
PVAMD_DEV void write_query(const MeshArgs& m, const QueryOut& out, int64_t i, V3 p, unsigned long long found, int hits) {
    const int f = (unsigned)(found >> 32) == 0x7F800000u ? -1 : (int)(unsigned)found;
    V3 q = v3(NAN, NAN, NAN);
    if (f >= 0) q = closest_on_record(m, m.rec_of_face[f], p);
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
    if (out.packed) {
        reinterpret_cast<f32x4*>(out.packed)[i] = f32x4{d, g.x, g.y, g.z};
    } else {
        out.dist[i] = d;
        out.grad[3 * i] = g.x;
        out.grad[3 * i + 1] = g.y;
        out.grad[3 * i + 2] = g.z;
    }
    if (out.face) out.face[i] = f;
    if (out.normal) {                            // :169-171
        out.normal[3 * i] = f >= 0 ? m.normal[3 * f] : NAN;
        out.normal[3 * i + 1] = f >= 0 ? m.normal[3 * f + 1] : NAN;
        out.normal[3 * i + 2] = f >= 0 ? m.normal[3 * f + 2] : NAN;
    }
}

// grid: x = groups of 64 points
// waves per SIMD the allocator is held to (it would settle for 6): 8 fit beside the LDS of an 8-wave block (C5 query with
// sign 7.4 -> 6.2 ms), 7 beside that of the smaller ones (2.16 M-point cache build 1.84 -> 1.73 ms; 8 spill: 1.85)
#ifndef PVAMD_MESH_MINWAVES
#define PVAMD_MESH_MINWAVES(SLICES) ((SLICES) == 8 ? 8 : ((SLICES) == 1 ? 6 : 7))
#endif
#ifndef PVAMD_CHAMFER_MINWAVES
#define PVAMD_CHAMFER_MINWAVES(SLICES) 1  /* the allocator's own choice (88 VGPRs, 5 waves): see profiles/r06_mesh_variants.txt */
#endif
template <int SLICES>
__global__ __launch_bounds__(64 * SLICES, PVAMD_MESH_MINWAVES(SLICES)) void mesh_query_kernel(MeshArgs m, const int* __restrict__ order,
                                                                const float* __restrict__ pts, int64_t P,
                                                                uint64_t seed, int64_t index_base, QueryOut out, HandOver ho) {
    __shared__ __attribute__((aligned(16))) MeshShared<SLICES, true> sh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const int64_t i = point_index(order, k, P);
    Wave<true> wv;
    wv.s.p = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (scan_mesh<SLICES, true>(m, sh, wv, wave, seed, index_base + i, ho, (int)blockIdx.x, 0)) return;
    if (wave != 0 || k >= P) return;
    write_query(m, out, i, wv.s.p, sh.g.best[lane], sh.g.hits[lane]);
}

// chamfer.py:81-82 transform_points, k-ordered fma chain
PVAMD_DEV V3 chamfer_point(const float* __restrict__ M, const float* __restrict__ pts, int64_t i) {
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    return v3(add_rn(fmaf(M[2], pz, fmaf(M[1], py, mul_rn(M[0], px))), M[3]),
              add_rn(fmaf(M[6], pz, fmaf(M[5], py, mul_rn(M[4], px))), M[7]),
              add_rn(fmaf(M[10], pz, fmaf(M[9], py, mul_rn(M[8], px))), M[11]));
}

// one wave's share of sum((scale * d)^2) from the winning (d2, face) of each of its points.  per == 0: the wave's points
// belong to ONE transform, whose sum is `sum`: a wave reduction and one atomic.  per > 0 (flat call: the points of all
// transforms, already transformed, in ONE spatial order): point i belongs to transform i / per -- one atomic per lane.
PVAMD_DEV void chamfer_accumulate(const MeshArgs& m, V3 p, bool live, unsigned long long found, float scale, double* __restrict__ sum,
                                  int64_t i = 0, int64_t per = 0) {
    const int best_f = (unsigned)(found >> 32) == 0x7F800000u ? -1 : (int)(unsigned)found;
    double acc = 0.0;
    if (live && best_f >= 0) {
        const V3 q = closest_on_record(m, m.rec_of_face[best_f], p);
        const float sd = mul_rn(scale, norm3_unfused(sub(q, p)));  // chamfer.py:92
        acc = (double)mul_rn(sd, sd);
    }
    if (per > 0) {
        if (live && best_f >= 0) atomicAdd(sum + i / per, acc);
        return;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

// grid: x = groups of 64 points, y = transform b (b0 = the slab's first transform).  W == nullptr: the flat call (see
// chamfer_accumulate) -- `pts` are the N = B * per transformed points of all transforms, y = 1.
template <int SLICES>
__global__ __launch_bounds__(64 * SLICES, PVAMD_CHAMFER_MINWAVES(SLICES)) void chamfer_mesh_kernel(MeshArgs m, const int* __restrict__ order,
                                                                  const float* __restrict__ W, int b0,
                                                                  const float* __restrict__ pts, int64_t N, float scale,
                                                                  double* __restrict__ out_sum, HandOver ho, int64_t per) {
    __shared__ __attribute__((aligned(16))) MeshShared<SLICES, false> sh;
    const int b = b0 + (int)blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const int64_t i = point_index(order, k, N);
    Wave<false> wv;
    wv.s.p = W ? chamfer_point(W + 16 * (int64_t)b, pts, i) : v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (scan_mesh<SLICES, false>(m, sh, wv, wave, 0, 0, ho, (int)blockIdx.x, b)) return;
    if (wave != 0) return;
    chamfer_accumulate(m, wv.s.p, k < N, sh.g.best[lane], scale, W ? out_sum + b : out_sum, i, W ? 0 : per);
}

// An upper bound of a point's distance to the mesh, by a greedy descent tile -> group -> record along the smallest
// |p - ctr| + r (every sphere contains whole triangles), and the exact distance to that record's triangle: a candidate like
// any other (`found` = (d^2, face), what the slots start from) and a bound that is a real distance instead of the far side
// of a sphere.  The blocks of the parts launch cannot hand each other their finds, so each would otherwise start from the
// tile-sphere bound (a tile radius too wide) and queue 3x the pairs.  Whole waves call this (v_readlane hands the tile
// spheres round); a NaN / inf point comes back with no finite bound.
PVAMD_DEV float greedy_bound(const MeshArgs& m, V3 p, unsigned long long& found) {
    const int lane = threadIdx.x & 63;
    float bound = INFINITY;
    found = kBestInit;
    if (m.F <= 0) return bound;
    const int ntiles = (m.F + kTile - 1) / kTile;
    const f32x4* spheres = reinterpret_cast<const f32x4*>(m.tiles);
    auto reach_of = [&](f32x4 sp) {
        const V3 w = v3(sp.x - p.x, sp.y - p.y, sp.z - p.z);
        return fast_sqrt(dot(w, w)) * 1.00001f + (sp.w + 1.1e-19f);  // slack: 1 ulp + the flushed denormals
    };
    int ti = 0;
    // every lane against every tile sphere: 64 spheres per vector load, handed round with v_readlane
    for (int base = 0; base < ntiles; base += 64) {
        const f32x4 mine = spheres[base + lane < ntiles ? base + lane : base];
        const int n = min(64, ntiles - base);
        for (int t = 0; t < n; ++t) {
            const f32x4 sp = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), t)),
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), t)),
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.z), t)),
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.w), t))};
            const float b = reach_of(sp);
            if (b < bound) { bound = b; ti = base + t; }
        }
    }
    int gi = 0;
    const int ngroups = (min(kTile, m.F - ti * kTile) + kGroup - 1) / kGroup;
    for (int k = 0; k < kGroupsPerTile; ++k) {
        if (k >= ngroups) break;
        const float b = reach_of(spheres[ntiles + ti * kGroupsPerTile + k]);
        if (b < bound) { bound = b; gi = k; }
    }
    const int j0 = ti * kTile + gi * kGroup;
    float nearest = INFINITY;
    int jn = -1;
    for (int k = 0; k < kGroup; ++k) {
        if (j0 + k >= m.F) break;
        const float b = reach_of(record_plane(m.rec, j0 + k, kPlaneSphere));
        if (b < nearest) { nearest = b; jn = j0 + k; }
    }
    bound = fminf(bound, nearest);
#ifndef PVAMD_MESH_NO_GREEDY_EXACT
    PVAMD_COUNT_PAIRS(m, 0, __popcll(__ballot(jn >= 0 && fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY)));
    if (jn >= 0 && fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY) {
        const f32x4 A = record_plane(m.rec, jn, kPlaneA), B = record_plane(m.rec, jn, kPlaneB), C = record_plane(m.rec, jn, kPlaneC);
        const V3 qp = sub(closest_point_triangle(p, xyz(A), xyz(B), xyz(C)), p);
        const float d2 = dot(qp, qp);
        found = ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(A.w);
        if (!(found < kBestInit)) found = kBestInit;  // a NaN d2
        else bound = fminf(bound, fast_sqrt(d2) * 1.00001f + 1.1e-19f);
    }
#endif
    return bound * 1.00001f;  // a NaN / inf point: never a finite bound
}

// ---- the listed groups: tiles spread over blocks ----------------------------------------------------------------
// A wave walks its flagged tiles one after the other; for a heavy group, or with only a few hundred point groups in
// the whole query, that serial walk -- not throughput -- sets the time.
//   list    few points: every group is listed up front (hand_over_all_kernel: rays, a bound and a first candidate per
//           point, the group's wave bound, its points in processing order); otherwise the scan above lists the heavy
//           ones as it finds them
//   parts   (nparts blocks of two or four waves per listed group)  block y takes the tiles ti % nparts == y, its waves
//           the 64-record passes of each; starts from the slot's bound; folds into the slots with global atomicMin /
//           atomicAdd.  Few points: the block that folds a group's last part in also writes the group's outputs
//   finish  (heavy groups only: one wave per listed group)  outputs from the slots
// (A `first` launch that visited the nearest tile and handed its bound on cost what it saved in the few-points path: C1
// 0.159 vs 0.158 ms, 1000 points 0.089 vs 0.063 ms without it.)
// block = one point group, two waves: wave 0 draws the rays, works out the group's bound and stores its points, wave 1
// works out the per-point bound (two independent serial chains: 5 + 8 us one after the other, 8 side by side)
__global__ __launch_bounds__(128) void hand_over_all_kernel(MeshArgs m, const int* __restrict__ order, const float* __restrict__ pts,
                                                            int64_t P, uint64_t seed, int64_t index_base, HandOver ho, int groups) {
    __shared__ float reach_of_lane[64];
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)g * 64 + lane;
    const int64_t i = point_index(order, slot, P);
    if (threadIdx.x < 64) {
        ho.hits[slot] = 0;
        const V3 dir = jitter_dir(m.ray_dir, seed, index_base + i);
        float* o = ho.dir + slot * 3;
        o[0] = dir.x; o[1] = dir.y; o[2] = dir.z;
        {   // the group's wave bound, once for all the blocks of the parts launch (same operations as scan_begin + wave_setup)
            Wave<true> wv{};
            wv.s.p = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
            const float inv_len = 1.f / sqrt_rn(dot(dir, dir));
            wv.dir = dir;
            wv.dn = v3(dir.x * inv_len, dir.y * inv_len, dir.z * inv_len);
            const bool any_live = wave_setup(m, wv);
            float* q = ho.pts + slot * 3;
            q[0] = wv.s.p.x; q[1] = wv.s.p.y; q[2] = wv.s.p.z;
            if (lane == 0) {
                ho.entries[2 * g] = g;
                ho.entries[2 * g + 1] = 0;
                if (g == 0) *ho.count = groups;
            }
            __syncthreads();  // the other wave's bounds
            if (any_live) {
                set_reach(wv.s, reach_of_lane[lane]);
                refresh_bound(wv);  // Q: what a block of the parts launch will start from
            }
            store_bound(ho.bound + (int64_t)g * kBoundFloats, wv.wb, any_live);
        }
        return;
    }
    unsigned long long found;
    const float bound = greedy_bound(m, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), found);
    ho.best[slot] = found;
    ho.reach[slot] = bound;
    reach_of_lane[lane] = bound;
    __syncthreads();
}
__global__ void hand_over_none_kernel(HandOver ho) { *ho.count = 0; }

// Few points that bring no processing order (pvamd_mesh_query_unordered, P <= 16384): the sort is one workgroup on one CU
// for ~16 us, and what the list launch works out per POINT -- the jittered ray, the greedy bound and the first candidate --
// does not need the order.  ONE launch: block 0 sorts, every other block takes 512 points in caller order (waves 0-7 the
// rays, waves 8-15 the bounds: two independent serial chains side by side) and stores by point index; the list launch that
// follows only gathers them into the slots and works out the groups' wave bounds.  C1: sort 15.8 + list 12.1 us ->
// 15.8 + 4.
constexpr int kPrepPoints = 512;
__global__ __launch_bounds__(1024) void mesh_small_prep_kernel(MeshArgs m, const float* __restrict__ pts, int P, uint64_t seed,
                                                               int64_t index_base, int* __restrict__ order, HandOver ho) {
    if (blockIdx.x == 0) {
        order_small_block(pts, P, order, nullptr, nullptr);
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = ((int)blockIdx.x - 1) * kPrepPoints + (wave & 7) * 64 + lane;
    const int ii = i < P ? i : P - 1;  // whole waves work (greedy_bound hands the tile spheres round the lanes)
    if (((int)blockIdx.x - 1) * kPrepPoints + (wave & 7) * 64 >= P) return;  // a wave past the end
    if (wave < 8) {
        const V3 dir = jitter_dir(m.ray_dir, seed, index_base + ii);
        if (i < P) {
            float* o = ho.p_dir + 3 * (int64_t)i;
            o[0] = dir.x; o[1] = dir.y; o[2] = dir.z;
        }
    } else {
        unsigned long long found;
        const float bound = greedy_bound(m, v3(pts[3 * ii], pts[3 * ii + 1], pts[3 * ii + 2]), found);
        if (i < P) {
            ho.p_best[i] = found;
            ho.p_reach[i] = bound;
        }
    }
}

// ... and the list launch of that path: one wave per group gathers its points' rays, bounds and candidates into the slot
// and works out the group's wave bound (what hand_over_all_kernel does in one go when the order comes with the call)
__global__ __launch_bounds__(64) void hand_over_gather_kernel(MeshArgs m, const int* __restrict__ order, const float* __restrict__ pts,
                                                              int64_t P, HandOver ho, int groups) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int64_t slot = (int64_t)g * 64 + lane;
    const int64_t i = point_index(order, slot, P);
    const V3 dir = v3(ho.p_dir[3 * i], ho.p_dir[3 * i + 1], ho.p_dir[3 * i + 2]);
    const float reach = ho.p_reach[i];
    ho.best[slot] = ho.p_best[i];
    ho.reach[slot] = reach;
    ho.hits[slot] = 0;
    float* o = ho.dir + slot * 3;
    o[0] = dir.x; o[1] = dir.y; o[2] = dir.z;
    Wave<true> wv{};
    wv.s.p = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    float* q = ho.pts + slot * 3;
    q[0] = wv.s.p.x; q[1] = wv.s.p.y; q[2] = wv.s.p.z;
    const float inv_len = 1.f / sqrt_rn(dot(dir, dir));
    wv.dir = dir;
    wv.dn = v3(dir.x * inv_len, dir.y * inv_len, dir.z * inv_len);
    const bool any_live = wave_setup(m, wv);
    if (any_live) {
        set_reach(wv.s, reach);
        refresh_bound(wv);
    }
    store_bound(ho.bound + (int64_t)g * kBoundFloats, wv.wb, any_live);
    if (lane == 0) {
        ho.entries[2 * g] = g;
        ho.entries[2 * g + 1] = 0;
        if (g == 0) *ho.count = groups;
    }
}

#ifndef PVAMD_MESH_PARTS_WAVES
#define PVAMD_MESH_PARTS_WAVES 7  // waves per SIMD the parts kernels are held to (C1: 0.165 ms at the allocator's 5, 0.144 at 6-7;
                                  // 30k points 0.201 / 0.190 / 0.207 ms at 6 / 7 / 8)
#endif
// one listed group in one block: wave w takes the w-th 64-record pass of the tiles ti % gridDim.y == blockIdx.y
template <bool WITH_RAY, bool WAIT = false, int WAVES = kTile / 64>
PVAMD_DEV void parts_of_group(const MeshArgs& m, MeshShared<WAVES, WITH_RAY>& sh, const int* __restrict__ order,
                              const float* __restrict__ M, const float* __restrict__ pts, int64_t P, uint64_t seed,
                              int64_t index_base, const HandOver& ho, int slot, int group) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!part_has_work<WITH_RAY>(m, ho.bound + (int64_t)slot * kBoundFloats, (int)blockIdx.y, (int)gridDim.y)) return;  // the whole block
    Wave<WITH_RAY> wv;
    // everything a block needs to start comes from the slot, in one batch of independent loads
    const float* q = ho.pts + ((int64_t)slot * 64 + lane) * 3;
    wv.s.p = v3(q[0], q[1], q[2]);
    const unsigned long long start = ho.best[(int64_t)slot * 64 + lane];
    const float known = ho.reach[(int64_t)slot * 64 + lane];
    TIC(t_parts);
    if (scan_begin(m, sh.g, wv, wave, seed, 0, &start, WITH_RAY ? ho.dir + (int64_t)slot * 192 : nullptr,
                   ho.bound + (int64_t)slot * kBoundFloats)) {
        // the bound worked out when the group was listed, or what it had found by then (pulled from the slots in scan_tiles);
        // the bound from the tile spheres only for a lane that has neither
        if (__any(wv.live && !(known < INFINITY) && (unsigned)(start >> 32) >= 0x7F800000u)) scan_seed(m, wv, 0, 1);
        if (known < wv.s.reach) set_reach(wv.s, known);
        refresh_bound(wv);
        constexpr int kPasses = (kTile / 64) / WAVES;  // 64-record passes of a tile per wave
        scan_tiles<WITH_RAY>(m, sh.g, sh.w[wave], wv, -1, (int)blockIdx.y, (int)gridDim.y, wave * kPasses, (wave + 1) * kPasses);
        scan_finish(m, sh.g, sh.w[wave], wv);
    }
    TOC(21, t_parts);
    __syncthreads();
    if (wave == 0) {
        if (WAIT) {  // returning atomics: when the old values are back, the updates have been performed
            unsigned long long was = 0ull;
            int had = 0;
            if (sh.g.best[lane] < start) was = atomicMin(&ho.best[(int64_t)slot * 64 + lane], sh.g.best[lane]);
            if (WITH_RAY && sh.g.hits[lane] != 0) had = atomicAdd(&ho.hits[(int64_t)slot * 64 + lane], sh.g.hits[lane]);
            asm volatile("" ::"v"(was), "v"(had) : "memory");
        } else {
            if (sh.g.best[lane] < start) atomicMin(&ho.best[(int64_t)slot * 64 + lane], sh.g.best[lane]);
            if (WITH_RAY && sh.g.hits[lane] != 0) atomicAdd(&ho.hits[(int64_t)slot * 64 + lane], sh.g.hits[lane]);
        }
    }
}

// few points: grid x = point groups (slot g = group g), y = part.  The block that folds a group's LAST part in writes the
// group's outputs (entries[2g + 1], zeroed by the list launch, counts the parts that are done): no finish launch.
// kAllWaves waves per block, each (kTile / 64) / kAllWaves of the passes of a tile
template <int kAllWaves>
__global__ __launch_bounds__(64 * kAllWaves, PVAMD_MESH_PARTS_WAVES) void mesh_parts_all_kernel(MeshArgs m, const int* __restrict__ order,
                                                                          const float* __restrict__ pts, int64_t P,
                                                                          uint64_t seed, int64_t index_base, HandOver ho, QueryOut out) {
    __shared__ __attribute__((aligned(16))) MeshShared<kAllWaves, true> sh;
    const int g = (int)blockIdx.x;
    parts_of_group<true, true, kAllWaves>(m, sh, order, nullptr, pts, P, seed, index_base, ho, g, g);
    if (threadIdx.x >= 64) return;
    // No fences: the slots are only ever touched by device-scope read-modify-write atomics inside this launch, this block's
    // have been performed (their old values are back) before its tick, and the block whose tick is the last reads the slots
    // with atomics as well.  (A __threadfence() per block -- an L2 write-back -- took the launch from 65 to 180 us.)
    const int lane = threadIdx.x;
    int done = 0;
    if (lane == 0) done = atomicAdd(&ho.entries[2 * g + 1], 1);
    if (__builtin_amdgcn_readfirstlane(done) != (int)gridDim.y - 1) return;
    const int64_t k = (int64_t)g * 64 + lane;
    if (k >= P) return;
    const int64_t i = order ? (int64_t)order[k] : k;
    const unsigned long long found = atomicMin(&ho.best[k], ~0ull);
    const int hits = atomicOr(&ho.hits[k], 0);
    write_query(m, out, i, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), found, hits);
}

// the listed groups: grid x = slots (strided over the list), y = part
template <bool WITH_RAY, int WAVES>
__global__ __launch_bounds__(64 * WAVES, PVAMD_MESH_PARTS_WAVES) void mesh_parts_kernel(MeshArgs m, const int* __restrict__ order,
                                                                      const float* __restrict__ W,
                                                                      const float* __restrict__ pts, int64_t P,
                                                                      uint64_t seed, int64_t index_base, HandOver ho) {
    __shared__ __attribute__((aligned(16))) MeshShared<WAVES, WITH_RAY> sh;
    const int listed = min(*ho.count, ho.cap);
    for (int slot = blockIdx.x; slot < listed; slot += gridDim.x) {
        parts_of_group<WITH_RAY, false, WAVES>(m, sh, order, W ? W + 16 * (int64_t)ho.entries[2 * slot + 1] : nullptr, pts, P, seed, index_base,
                                 ho, slot, ho.entries[2 * slot]);
        __syncthreads();  // the slots in LDS are reused by the next listed group
    }
}

__global__ __launch_bounds__(64) void mesh_query_finish_kernel(MeshArgs m, const int* __restrict__ order,
                                                               const float* __restrict__ pts, int64_t P, HandOver ho,
                                                               QueryOut out) {
    const int listed = min(*ho.count, ho.cap);
    for (int slot = blockIdx.x; slot < listed; slot += gridDim.x) {
        const int64_t k = (int64_t)ho.entries[2 * slot] * 64 + threadIdx.x;
        if (k >= P) continue;
        const int64_t i = order ? (int64_t)order[k] : k;
        write_query(m, out, i, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), ho.best[(int64_t)slot * 64 + threadIdx.x],
                    ho.hits[(int64_t)slot * 64 + threadIdx.x]);
    }
}

__global__ __launch_bounds__(64) void chamfer_finish_kernel(MeshArgs m, const int* __restrict__ order, const float* __restrict__ W,
                                                            const float* __restrict__ pts, int64_t N, float scale, HandOver ho,
                                                            double* __restrict__ out_sum, int64_t per) {
    const int listed = min(*ho.count, ho.cap);
    for (int slot = blockIdx.x; slot < listed; slot += gridDim.x) {
        const int64_t k = (int64_t)ho.entries[2 * slot] * 64 + threadIdx.x;
        const int b = ho.entries[2 * slot + 1];
        const int64_t i = point_index(order, k, N);
        const V3 p = W ? chamfer_point(W + 16 * (int64_t)b, pts, i) : v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        chamfer_accumulate(m, p, k < N, ho.best[(int64_t)slot * 64 + threadIdx.x], scale, W ? out_sum + b : out_sum, i, W ? 0 : per);
    }
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
    if (k < F) rec_of_face[__float_as_int(record_plane(rec, k, kPlaneA).w)] = k;
}

static MeshArgs mesh_args(const pvamd_mesh_t& mesh) {
    MeshArgs m;
    m.normal = mesh.normal;
    m.rec = mesh.rec;
    m.tiles = mesh.tiles;
    m.rec_of_face = mesh.rec_of_face;
    m.F = mesh.F;
    for (int d = 0; d < 3; ++d) m.ray_dir[d] = mesh.ray_dir[d];
    m.pairs = reinterpret_cast<unsigned long long*>(mesh.pair_counters);
    return m;
}

constexpr int kMaxFaces = 1 << 26;  // queue entries are record << 6 | lane
#ifndef PVAMD_MESH_MAX_SLICES
#define PVAMD_MESH_MAX_SLICES 8
#endif
#ifndef PVAMD_MESH_MIN_PARTS
#define PVAMD_MESH_MIN_PARTS 4
#endif
constexpr int kMinParts = PVAMD_MESH_MIN_PARTS;     // fewer parts than this (a mesh of less than ~12 tiles): the single launch

// How many waves share one 64-point group (every wave needs tiles of its own: ti % slices == wave).
//   many groups           -> 2: the least replicated per-wave work (2 M points on the 62-tile drill: 1.9 ms with 2, 2.05
//                            with 4, 2.4 with 1, 3.1 with 8);
//   fewer groups          -> 4, 8, so that the 1024 SIMDs still fill (100k points on the drill: 0.40 ms with 8, 0.47 with
//                            4, 0.79 with 2);
//   many tiles            -> the work per group is heavy-tailed (a point near the medial axis is equidistant to much of
//                            the surface and needs most tiles) and the slowest groups set the kernel time: 8 when nothing
//                            can be handed over (C5, 389 tiles: 5.2 ms with 8, 11.5 with 2); when the heavy groups go
//                            to a launch of their own the tail is gone and the count of groups decides as above (C5:
//                            3.17 ms with 2, 3.55 with 4).
// Voxel centres of a regular grid (the cartesian product of three coordinate arrays, x slowest: voxel.py:20-25) and a
// processing order for them in ONE launch: thread i writes centre i (caller order = the cache's C order) and its position in an
// order that walks the grid in 4 x 4 x 4 bricks (clipped at the far faces) -- so that a run of 64 consecutive positions is a cube
// of 64 neighbouring centres, the most compact group the mesh kernels can be given -- with no sort: the position of a voxel
// follows from its coordinates in closed form.  (One level of bricks in plain C order was 10 % slower on a 2.2 M-voxel grid than
// the Hilbert sort it replaces; with the bricks themselves grouped in 16^3 super-bricks it is level or ahead at every size:
// drill 37x33x40 0.29 -> 0.27 ms, 132x113x145 1.48 -> 1.33 ms, wrench 218x126x111 1.05 -> 1.03 ms, tools/build_probe*.py.)
__global__ __launch_bounds__(256) void grid_points_kernel(const float* __restrict__ cx, const float* __restrict__ cy,
                                                          const float* __restrict__ cz, int nx, int ny, int nz,
                                                          float* __restrict__ pts, int* __restrict__ order) {
    const int64_t n = (int64_t)nx * ny * nz;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int z = (int)(i % nz), y = (int)((i / nz) % ny), x = (int)(i / ((int64_t)nz * ny));
    pts[3 * i] = cx[x];
    pts[3 * i + 1] = cy[y];
    pts[3 * i + 2] = cz[z];
    // two levels: 16^3 super-bricks in C order, 4^3 bricks in C order inside a super-brick, voxels in C order inside a brick (each
    // level clipped at the far faces), so that the 64 groups of a super-brick -- neighbours in space -- are neighbours in the launch
    // too and share the mesh tiles they pull through the caches.  "Voxels before this one" at every level is the same closed form:
    // the slabs before it (all full), the rows before it inside its slab, the boxes before it inside its row.
    const int Bx = x >> 4, By = y >> 4, Bz = z >> 4;
    const int Wx = min(16, nx - 16 * Bx), Wy = min(16, ny - 16 * By), Wz = min(16, nz - 16 * Bz);  // the super-brick's extents
    const int sx = x & 15, sy = y & 15, sz = z & 15;                                                  // position inside it
    const int bx = sx >> 2, by = sy >> 2, bz = sz >> 2;
    const int wx = min(4, Wx - 4 * bx), wy = min(4, Wy - 4 * by), wz = min(4, Wz - 4 * bz);        // the brick's extents
    const int64_t j = (int64_t)(16 * Bx) * ny * nz + (int64_t)Wx * (16 * By) * nz + (int64_t)Wx * Wy * (16 * Bz) +
                      ((int64_t)(4 * bx) * Wy * Wz + (int64_t)wx * (4 * by) * Wz + (int64_t)wx * wy * (4 * bz)) +
                      ((int64_t)((sx & 3) * wy + (sy & 3)) * wz + (sz & 3));
    order[j] = (int)i;
}

static int pick_slices(int64_t groups, int mesh_tiles, bool hand_over) {
    int s = 2;
    while (s < 8 && (int64_t)s * groups < 16384) s <<= 1;
    if (mesh_tiles > 128 && !hand_over) s = 8;
    if (s > PVAMD_MESH_MAX_SLICES) s = PVAMD_MESH_MAX_SLICES;
    while (s > 1 && s > mesh_tiles) s >>= 1;
    return s;
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
    hipLaunchKernelGGL(mesh_prepare_records, dim3((F + kTile - 1) / kTile), dim3(kTile), 0, s, tri, face_id, F, abs_margin, rec_out);
    hipLaunchKernelGGL(mesh_prepare_tiles, dim3((F + kTile - 1) / kTile), dim3(256), 0, s, rec_out, F, abs_margin, tiles_out);
    hipLaunchKernelGGL(invert_face_order_kernel, dim3((F + 255) / 256), dim3(256), 0, s, rec_out, F, rec_of_face_out);
    return (int)hipGetLastError();
}

// grid.x of the launches that walk the list of handed-over groups
static unsigned list_blocks(int cap) { return (unsigned)(cap < 512 ? (cap < 1 ? 1 : cap) : 512); }

// the parts launch over the listed (heavy) groups: x = slots (a block strides over the list), y = parts
template <bool WITH_RAY>
static void launch_heavy_parts(const MeshArgs& m, const int* order, const float* W, const float* points, int64_t P, uint64_t seed,
                               int64_t index_base, const HandOver& ho, int ntiles, hipStream_t s) {
    int parts = kHeavyParts, waves = 4;
    unsigned xb = list_blocks(ho.cap);
#ifdef PVAMD_MESH_TUNE
    if (getenv("PVAMD_TUNE_HPARTS")) parts = atoi(getenv("PVAMD_TUNE_HPARTS"));
    if (getenv("PVAMD_TUNE_HWAVES")) waves = atoi(getenv("PVAMD_TUNE_HWAVES"));
    if (getenv("PVAMD_TUNE_HBLOCKS")) xb = (unsigned)atoi(getenv("PVAMD_TUNE_HBLOCKS"));
    if (xb > (unsigned)ho.cap) xb = (unsigned)ho.cap;
#endif
    if (parts > ntiles) parts = ntiles;
    const dim3 grid(xb, (unsigned)parts);
    if (waves == 2) hipLaunchKernelGGL((mesh_parts_kernel<WITH_RAY, 2>), grid, dim3(128), 0, s, m, order, W, points, P, seed, index_base, ho);
    else hipLaunchKernelGGL((mesh_parts_kernel<WITH_RAY, 4>), grid, dim3(256), 0, s, m, order, W, points, P, seed, index_base, ho);
}

// sort_into != nullptr: the caller brings no order; one is worked out into sort_into (P <= kSmallPoints)
static int mesh_query_impl(const pvamd_mesh_t* mesh, const float* points, const int32_t* order, int32_t* sort_into, int64_t P,
                           uint64_t jitter_seed, int64_t index_base, float* out_closest, float* out_dist, float* out_grad,
                           int32_t* out_face, float* out_normal, void* scratch, void* stream, float* out_packed = nullptr) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!mesh || (!out_packed && (!out_dist || !out_grad))) return PVAMD_E_NULL;
    if (mesh->F < 0) return PVAMD_E_SHAPE;
    if (!points || (mesh->F > 0 && (!mesh->rec || !mesh->tiles || !mesh->normal || !mesh->rec_of_face))) return PVAMD_E_NULL;
    if (scratch && !aligned_to(scratch, 8)) return PVAMD_E_ALIGN;
    if (mesh->F > kMaxFaces) return PVAMD_E_SHAPE;
    const MeshArgs m = mesh_args(*mesh);
    const int64_t groups = (P + 63) / 64;
    if (groups > 0x7fffffff) return PVAMD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const QueryOut out{out_closest, out_dist, out_grad, out_face, out_normal, out_packed};
    const int ntiles = (mesh->F + kTile - 1) / kTile;
    
    const int cap = (int)PVAMD_MESH_SCRATCH_SLOTS(P);  // what PVAMD_MESH_SCRATCH_BYTES(P) holds
    const HandOver ho = hand_over(scratch, cap);
    const int slices = pick_slices(groups, ntiles, ho.cap > 0 && ntiles >= kHeavyMinTiles);
    // few point groups, many tiles: spread each group's tiles over `parts` blocks of `aw` waves (see mesh_parts_kernel)
    // Sweep over 1k .. 520k points x parts x waves on the 62-tile drill and the 389-tile sphere (profiles/r04_mesh_variants.txt,
    // section 7): tiles per block by the number of point groups whatever the size of the mesh -- one for up to 64 groups, one
    // and a half up to 256, three up to ~2000, then a five-hundredth of the groups --, two waves per block instead of four once the
    // launch is beyond ~100k waves.
    // (in halves of a tile)
    const int half_tiles = groups <= 64 ? 2 : (groups <= 256 ? 3 : (groups < 2048 ? 6 : (int)(groups / 256)));
    int parts = (2 * ntiles + half_tiles - 1) / half_tiles;
    if (parts > 65535) parts = 65535;  // gridDim.y (a mesh of more than 16.7 M triangles)
    int aw = (int64_t)groups * parts * 4 > 100000 ? 2 : 4;
    bool few = parts >= kMinParts;
#ifdef PVAMD_MESH_TUNE
    if (getenv("PVAMD_TUNE_PARTS")) {  // parts = 0: the single launch
        parts = atoi(getenv("PVAMD_TUNE_PARTS"));
        aw = getenv("PVAMD_TUNE_WAVES") ? atoi(getenv("PVAMD_TUNE_WAVES")) : 4;
        few = parts > 0;
        if (parts > ntiles) parts = ntiles;
    }
#endif
    const bool two_launches = ho.cap > 0 && groups <= ho.cap && few;
    if (sort_into) {
        order = sort_into;
        if (two_launches) {  // the sort in block 0 of the launch that works the rays and bounds out per point
            hipLaunchKernelGGL(mesh_small_prep_kernel, dim3(1 + (unsigned)((P + kPrepPoints - 1) / kPrepPoints)), dim3(1024), 0, s, m,
                               points, (int)P, jitter_seed, index_base, sort_into, ho);
            hipLaunchKernelGGL(hand_over_gather_kernel, dim3((unsigned)groups), dim3(64), 0, s, m, order, points, P, ho, (int)groups);
        } else {
            const int rc = pvamd_morton_order(points, P, sort_into, nullptr, nullptr, sort_into, stream);  // (one launch: no scratch used)
            if (rc != 0) return rc;
        }
    }
    if (two_launches) {
        if (!sort_into)
            hipLaunchKernelGGL(hand_over_all_kernel, dim3((unsigned)groups), dim3(128), 0, s, m, order, points, P, jitter_seed,
                               index_base, ho, (int)groups);
        const dim3 grid((unsigned)groups, (unsigned)parts);
        switch (aw) {
            case 2: hipLaunchKernelGGL((mesh_parts_all_kernel<2>), grid, dim3(128), 0, s, m, order, points, P, jitter_seed, index_base, ho, out); break;
            default: hipLaunchKernelGGL((mesh_parts_all_kernel<4>), grid, dim3(256), 0, s, m, order, points, P, jitter_seed, index_base, ho, out); break;
        }
        return (int)hipGetLastError();
    }
    const bool heavy = ho.cap > 0 && ntiles >= kHeavyMinTiles;
    const HandOver none = hand_over(nullptr, 0);
    if (heavy) hipLaunchKernelGGL(hand_over_none_kernel, dim3(1), dim3(1), 0, s, ho);
    switch (slices) {
        case 8: hipLaunchKernelGGL((mesh_query_kernel<8>), dim3((unsigned)groups), dim3(512), 0, s, m, order, points, P, jitter_seed, index_base, out, heavy ? ho : none); break;
        case 4: hipLaunchKernelGGL((mesh_query_kernel<4>), dim3((unsigned)groups), dim3(256), 0, s, m, order, points, P, jitter_seed, index_base, out, heavy ? ho : none); break;
        case 2: hipLaunchKernelGGL((mesh_query_kernel<2>), dim3((unsigned)groups), dim3(128), 0, s, m, order, points, P, jitter_seed, index_base, out, heavy ? ho : none); break;
        default: hipLaunchKernelGGL((mesh_query_kernel<1>), dim3((unsigned)groups), dim3(64), 0, s, m, order, points, P, jitter_seed, index_base, out, heavy ? ho : none); break;
    }
    if (heavy) {
        launch_heavy_parts<true>(m, order, nullptr, points, P, jitter_seed, index_base, ho, ntiles, s);
        hipLaunchKernelGGL(mesh_query_finish_kernel, dim3(list_blocks(ho.cap)), dim3(64), 0, s, m, order, points, P, ho, out);
    }
    return (int)hipGetLastError();
}

extern "C" int pvamd_mesh_query(const pvamd_mesh_t* mesh, const float* points, const int32_t* order, int64_t P,
                                uint64_t jitter_seed, int64_t index_base, float* out_closest, float* out_dist, float* out_grad,
                                int32_t* out_face, float* out_normal, void* scratch, void* stream) {
    return mesh_query_impl(mesh, points, order, nullptr, P, jitter_seed, index_base, out_closest, out_dist, out_grad, out_face,
                           out_normal, scratch, stream);
}

extern "C" int pvamd_mesh_query_unordered(const pvamd_mesh_t* mesh, const float* points, int64_t P, uint64_t jitter_seed,
                                          int64_t index_base, float* out_closest, float* out_dist, float* out_grad,
                                          int32_t* out_face, float* out_normal, int32_t* order_scratch, void* scratch, void* stream) {
    if (P > kSmallPoints) return PVAMD_E_SHAPE;
    if (P > 0 && !order_scratch) return PVAMD_E_NULL;
    return mesh_query_impl(mesh, points, nullptr, order_scratch, P, jitter_seed, index_base, out_closest, out_dist, out_grad,
                           out_face, out_normal, scratch, stream);
}

extern "C" int pvamd_cache_build(const pvamd_mesh_t* mesh, const float* cx, const float* cy, const float* cz, int32_t nx, int32_t ny,
                                 int32_t nz, uint64_t jitter_seed, float* out_packed, float* points_scratch, int32_t* order_scratch,
                                 void* scratch, void* stream) {
    if (nx < 1 || ny < 1 || nz < 1 || (int64_t)nx * ny * nz > (int64_t)INT32_MAX) return PVAMD_E_SHAPE;
    if (!mesh || !cx || !cy || !cz || !out_packed || !points_scratch || !order_scratch) return PVAMD_E_NULL;
    if (!aligned_to(out_packed, 16)) return PVAMD_E_ALIGN;
    const int64_t n = (int64_t)nx * ny * nz;
    hipLaunchKernelGGL(grid_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cx, cy, cz, nx, ny, nz,
                       points_scratch, order_scratch);
    return mesh_query_impl(mesh, points_scratch, order_scratch, nullptr, n, jitter_seed, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                           scratch, stream, out_packed);
}

// W != nullptr: B transforms x N points, grid (groups, B).  W == nullptr: the flat call -- N = B * per transformed points.
static int launch_chamfer_mesh(const pvamd_mesh_t* mesh, const float* W, int32_t B, const float* points, const int32_t* order,
                               int64_t N, int64_t per, float scale, double* out_sum, void* scratch, void* stream) {
    if (B < 0 || N < 0) return PVAMD_E_SHAPE;
    if (B == 0) return 0;
    if (!mesh || !out_sum) return PVAMD_E_NULL;
    if (mesh->F < 0) return PVAMD_E_SHAPE;
    if (scratch && !aligned_to(scratch, 8)) return PVAMD_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_f64_kernel, dim3((B + 255) / 256), dim3(256), 0, s, out_sum, B);
    if (N == 0 || mesh->F == 0) return (int)hipGetLastError();
    if (!points || !mesh->rec || !mesh->tiles || !mesh->rec_of_face) return PVAMD_E_NULL;
    if (mesh->F > kMaxFaces) return PVAMD_E_SHAPE;
    const MeshArgs m = mesh_args(*mesh);
    const int64_t groups = (N + 63) / 64;
    if (groups > 0x7fffffff) return PVAMD_E_SHAPE;
    const int ntiles = (mesh->F + kTile - 1) / kTile;
    const int cap = (int)PVAMD_MESH_SCRATCH_SLOTS(N);
    const HandOver ho = hand_over(ntiles >= kHeavyMinTiles ? scratch : nullptr, cap);
    const int32_t ny = W ? B : 1;
    const int slices = pick_slices(groups * (int64_t)ny, ntiles, ho.cap > 0);
    if (ho.cap > 0) hipLaunchKernelGGL(hand_over_none_kernel, dim3(1), dim3(1), 0, s, ho);
    // y-dimension of a HIP grid is limited to 65535: walk B in slabs
    for (int32_t b0 = 0; b0 < ny; b0 += 65535) {
        const int32_t nb = (ny - b0) < 65535 ? (ny - b0) : 65535;
        const dim3 grid((unsigned)groups, nb);
        switch (slices) {
            case 8: hipLaunchKernelGGL((chamfer_mesh_kernel<8>), grid, dim3(512), 0, s, m, order, W, b0, points, N, scale, out_sum, ho, per); break;
            case 4: hipLaunchKernelGGL((chamfer_mesh_kernel<4>), grid, dim3(256), 0, s, m, order, W, b0, points, N, scale, out_sum, ho, per); break;
            case 2: hipLaunchKernelGGL((chamfer_mesh_kernel<2>), grid, dim3(128), 0, s, m, order, W, b0, points, N, scale, out_sum, ho, per); break;
            default: hipLaunchKernelGGL((chamfer_mesh_kernel<1>), grid, dim3(64), 0, s, m, order, W, b0, points, N, scale, out_sum, ho, per); break;
        }
    }
    if (ho.cap > 0) {
        launch_heavy_parts<false>(m, order, W, points, N, 0, 0, ho, ntiles, s);
        hipLaunchKernelGGL(chamfer_finish_kernel, dim3(list_blocks(ho.cap)), dim3(64), 0, s, m, order, W, points, N, scale, ho, out_sum, per);
    }
    return (int)hipGetLastError();
}

extern "C" int pvamd_chamfer_mesh(const pvamd_mesh_t* mesh, const float* W, int32_t B, const float* points,
                                  const int32_t* order, int64_t N, float scale, double* out_sum, void* scratch, void* stream) {
    if (B > 0 && N > 0 && !W) return PVAMD_E_NULL;
    return launch_chamfer_mesh(mesh, W, B, points, order, N, 0, scale, out_sum, scratch, stream);
}

extern "C" int pvamd_chamfer_mesh_flat(const pvamd_mesh_t* mesh, int32_t B, const float* points, const int32_t* order,
                                       int64_t per, float scale, double* out_sum, void* scratch, void* stream) {
    if (per < 0) return PVAMD_E_SHAPE;
    return launch_chamfer_mesh(mesh, nullptr, B, points, order, (int64_t)B * per, per, scale, out_sum, scratch, stream);
}
