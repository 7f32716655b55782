/*
 * pvamd.h -- C ABI of libpvamd.so, the MI355X (gfx950) batched SDF query engine.
 *
 * The reference (UM-ARM-Lab/pytorch_volumetric @ 0.5.2) has no FFI: its boundary is the Python
 * ObjectFrameSDF protocol (sdf.py:217-246).  These entry points are what the Python classes in
 * pytorch_volumetric_amd/ bind with ctypes; each one replaces a *sequence* of stock torch ops /
 * Embree calls in the reference, cited per function below (paths relative to
 * src/pytorch_volumetric/ in the reference).
 *
 * Conventions (all entry points):
 *   - extern "C"; return 0 on success, <0 = invalid argument (PVAMD_E_*), >0 = hipError_t.
 *   - every pointer documented "device" is a device (HBM) address; "host" is ordinary memory.
 *   - fp32 buffers are dense row-major; points are AoS [P][3].
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and the call returns without
 *     synchronising; nothing is allocated or freed; no static mutable state (re-entrant).
 *   - 4x4 transforms are row-major, column-vector convention: x' = M[:3,:3] x + M[:3,3]
 *     (chamfer.py:14, tests/test_model_to_sdf.py:277-279).
 */
#ifndef PVAMD_H
#define PVAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVAMD_ABI_VERSION 12

#define PVAMD_E_NULL      (-1)  /* a required pointer is NULL            */
#define PVAMD_E_SHAPE     (-2)  /* a size/shape argument is out of range */
#define PVAMD_E_ALIGN     (-3)  /* a pointer is not 4-byte aligned       */
#define PVAMD_E_MODE      (-4)  /* unknown enum value                    */

/* sdf.py:436-438 OutOfBoundsStrategy */
#define PVAMD_OOB_LOOKUP_GT_SDF 0   /* kernel writes zeros + sets out_oob; caller queries gt_sdf on that subset */
#define PVAMD_OOB_BOUNDING_BOX  1   /* distance to the (unpadded) surface bounding box, sdf.py:555-571          */

/*
 * One cached voxel grid = the read-only state of a CachedSDF (sdf.py:441-525) plus the value-range
 * view the reference wraps around it (multidim_indexing TorchMultidimView, constructed sdf.py:521).
 *
 * HBM layout: ONE 16-byte record per voxel, vox[flat] = (val, gx, gy, gz), flat = (kx*ny + ky)*nz + kz
 * (C order: x slowest, z fastest -- voxel.py:20-25 cartesian_prod, sdf.py:504-505).  The reference keeps
 * two arrays (val [nx,ny,nz] and grad [n,3]); packing makes a query one 16-B gather.
 *
 * Index arithmetic is carried out in the dtype the reference's torch promotion would use:
 * index_f64 = 1 when the value range reached the view as float64 (numpy ranges, the README flow),
 * 0 when it reached it as float32 (python-float ranges).  Both triples are always filled in.
 *
 * `rule`: the choices of the view that NO reference test pins (multidim_indexing is an un-vendored, un-pinned
 * dependency; call sites sdf.py:521,537-540).  0 = what this library restates by default: the index is
 * round-half-to-even((p - min) / res), a point is valid when min <= p <= max (tested on the VALUE).  The bits
 * select the alternatives, so that matching the real package is a flag, not a kernel change:
 *   PVAMD_RULE_VALID_ON_INDEX   valid when 0 <= index < shape on the ROUNDED index (NaN / infinite quotients: invalid),
 *                               i.e. the range grows by half a voxel on every side
 *   PVAMD_RULE_ROUND_HALF_AWAY  index = round((p - min) / res), halves away from zero (C roundf / round)
 *   PVAMD_RULE_ROUND_FLOOR_HALF index = floor((p - min) / res + 0.5), the sum rounded in the index dtype
 *   PVAMD_RULE_RES_F64          (host side; nothing in the kernels reads it) a float32 range's resolution was evaluated
 *                               in float64 and then rounded to float32, instead of in float32 throughout
 * The query kernels' fast paths are rule-independent: their range test compares with vlo / vhi and their index
 * estimate falls back to the exact statement near every half-integer -- pvamd_grid_finalize() derives vlo / vhi for the
 * rule, and the exact statements (grid_lookup.h voxel_index_1d) follow it.
 */
#define PVAMD_RULE_VALID_ON_INDEX   1
#define PVAMD_RULE_ROUND_HALF_AWAY  2
#define PVAMD_RULE_ROUND_FLOOR_HALF 4
#define PVAMD_RULE_RES_F64          8
typedef struct pvamd_grid {
    const float* vox;        /* device, [nx*ny*nz][4]                                              */
    double       dmin[3];    /* range minimum per dim (float64 view)                               */
    double       dmax[3];    /* range maximum per dim                                              */
    double       dres[3];    /* (dmax-dmin)/(shape-1) evaluated in float64                         */
    float        fmin[3];    /* float32(range min)                                                 */
    float        fmax[3];    /* float32(range max)                                                 */
    float        fres[3];    /* (fmax-fmin)/float32(shape-1) evaluated in float32                  */
    float        bb_min[3];  /* gt_sdf.surface_bounding_box()[:,0], NO padding (sdf.py:525)        */
    float        bb_max[3];  /* ...[:,1]                                                           */
    int32_t      shape[3];   /* nx, ny, nz (each >= 2)                                             */
    int32_t      index_f64;  /* see above                                                          */
    int32_t      oob_mode;   /* PVAMD_OOB_*                                                        */
    int32_t      finalized;  /* set by pvamd_grid_finalize(); kernels refuse descriptors without it */
    /* ---- derived by pvamd_grid_finalize() from the fields above; callers do not fill these ---- */
    float        vlo[3];     /* smallest float32 p that is valid under `rule` (exact range test in fp32)       */
    float        vhi[3];     /* largest  float32 p that is valid                                             */
    float        inv32[3];   /* float32(1/res): the multiply-first index estimate, checked against a rounding */
    float        err32[3];   /* bound and redone with the exact IEEE division when it is within it            */
    /* ---- filled by the caller (before pvamd_grid_finalize) ---- */
    int32_t      rule;       /* PVAMD_RULE_* bits; 0 = the default restatement (occupies former padding)      */
    /* ---- float64 query points (the *_f64 entry points) ---- */
    double       dbb_min[3]; /* surface bounding box in float64: sdf.py:556-557 casts self.bb to the query dtype   */
    double       dbb_max[3];
    /* ---- derived by pvamd_grid_finalize() (ABI 11) ---- */
    float        range_n2;   /* the largest squared bounding-box distance fma(tz,tz,fma(ty,ty,tx*tx)) (sdf.py:559-568, float32)
                                any VALID p can have: the statements are monotone, so it is their value at the corner of
                                [vlo, vhi] farthest from the box.  n2 > range_n2 proves "out of range" with one compare.
                                +inf when the descriptor has no box (NaN bounds).                                   */
    float        reserved0;
} pvamd_grid_t;

/*
 * One triangle mesh = the read-only state of an ObjectFactory after precompute_sdf (sdf.py:97-120), prepared for the
 * kernels by pvamd_mesh_prepare().
 * normal:  device [F][3] fp32 unit face normals (sdf.py:119-120), indexed by ORIGINAL face id
 * rec:     device float[PVAMD_REC_FLOATS(F)] per-triangle records written by pvamd_mesh_prepare, in the (spatially
 *          sorted) order the caller chose, stored in whole tiles of PVAMD_TRI_TILE records (opaque layout: six float4
 *          planes per tile); each record carries its original face id
 * tiles:   device float[PVAMD_TILES_FLOATS(F)]: [T][4] bounding sphere (cx, cy, cz, r) of each run of PVAMD_TRI_TILE
 *          records (T = ceil(F/PVAMD_TRI_TILE)), followed by [T][PVAMD_TRI_TILE/PVAMD_TRI_GROUP][4] spheres of each run
 *          of PVAMD_TRI_GROUP records
 * rec_of_face: device [F] int32, position of the record of original face id f (inverse of the chosen order)
 * ray_dir: the reference passes bounding_box(padding=1.0)[:,1] as the last three floats of each ray
 *          (sdf.py:147-152); open3d reads those as the ray DIRECTION (tnear=0, tfar=inf).  Kept as float64
 *          because the reference adds its jitter in float64 before rounding to float32 (sdf.py:149-150).
 */
#define PVAMD_TRI_REC   24   /* floats per prepared triangle record */
#define PVAMD_TRI_TILE  256  /* triangles per tile / per tile sphere */
#define PVAMD_TRI_GROUP 16   /* triangles per group sphere */
#define PVAMD_REC_FLOATS(F) ((((F) + PVAMD_TRI_TILE - 1) / PVAMD_TRI_TILE) * PVAMD_TRI_TILE * PVAMD_TRI_REC)
#define PVAMD_TILES_FLOATS(F) ((((F) + PVAMD_TRI_TILE - 1) / PVAMD_TRI_TILE) * (4 + 4 * (PVAMD_TRI_TILE / PVAMD_TRI_GROUP)))
typedef struct pvamd_mesh {
    const float* normal;     /* device */
    const float* rec;        /* device */
    const float* tiles;      /* device */
    const int32_t* rec_of_face; /* device */
    int32_t      F;
    int32_t      reserved;
    double       ray_dir[3];
    uint64_t*    pair_counters; /* device, NULL (the default) or two counters the mesh kernels add to: exact closest-point tests
                                   and exact ray tests executed (measurement only: what the broad phase left to do)            */
} pvamd_mesh_t;

/* One frame of a kinematic tree (URDF link + the joint that attaches it to its parent), frames sorted parents-first. */
typedef struct pvamd_joint {
    int32_t parent;      /* index of the parent frame, -1 for the root                                       */
    int32_t jtype;       /* 0 fixed, 1 revolute / continuous, 2 prismatic                                    */
    int32_t jcol;        /* column of q that drives this joint (ignored for fixed joints)                    */
    int32_t leaf_slot;   /* s >= 0: also write this frame's world matrix to link_world_out[s*A + a]; -1: no  */
    float   axis[3];     /* unit joint axis in the joint frame                                               */
    float   reserved;
    float   origin[12];  /* rows 0..2 of the parent-link -> joint-frame transform (URDF <origin>)            */
} pvamd_joint_t;

int         pvamd_abi_version(void);
const char* pvamd_build_info(void);      /* static string: arch, compiler, build flags */
/* number of devices visible / name of device 0 -- lets a host language check the GPU without a HIP binding */
int         pvamd_device_count(void);

/* Host-side, pure: fill the derived fields of a descriptor (vlo/vhi/inv32/err32, finalized) from its primary
 * fields.  Must be called once after the primary fields are set and before the descriptor is used or uploaded.   */
int pvamd_grid_finalize(pvamd_grid_t* grid);

/* Build the packed voxel layout on device: out[i] = (val[i], grad[i][0..2]).  Replaces nothing in the
 * reference (layout choice); feeds every grid entry point.  val: device [n], grad: device [n][3], out: device [n][4]. */
int pvamd_pack_grid(const float* val, const float* grad, int64_t n, float* out, void* stream);

/* CachedSDF.__call__ (sdf.py:535-571): nearest-voxel lookup of (val, grad) with out-of-bounds handling.
 * grid: host.  points: device [P][3].  out_val: device [P].  out_grad: device [P][3].
 * out_oob: device [P] bytes or NULL; 1 where the point failed the range test (sdf.py:540-541).        */
int pvamd_cached_query(const pvamd_grid_t* grid, const float* points, int64_t P,
                       float* out_val, float* out_grad, uint8_t* out_oob, void* stream);

/* Host-side, pure: which kernel pvamd_cached_query launches for P points (one launch for any P; the choice depends on P
 * only) -- for tests that want every kernel covered and for a profile that wants to name the kernel it measured.        */
#define PVAMD_CQ_KERNEL_SCALAR     0  /* one point per lane, grid-stride                                  (P < 16,384) */
#define PVAMD_CQ_KERNEL_DIRECT_1   1  /* cached_query_direct: no LDS, 1 point per lane,  8 waves per workgroup        */
#define PVAMD_CQ_KERNEL_DIRECT_2   2  /*                              2 points per lane, 4 waves                      */
#define PVAMD_CQ_KERNEL_DIRECT_2W  3  /*                              2 points per lane, 16 waves (about 1M points)   */
#define PVAMD_CQ_KERNEL_DIRECT_4   4  /*                              4 points per lane, 4 waves                      */
#define PVAMD_CQ_KERNEL_WAVE_TILE  5  /* cached_query_wave: 256-point tiles through LDS                               */
#define PVAMD_CQ_KERNEL_STREAMING  6  /* cached_query_wave, streaming instantiation                      (P > 8M)    */
int pvamd_cached_query_kernel(int64_t P);

/* CachedSDF.outside_surface (sdf.py:593-602): OOB -> 1, else vox.val > level.  out: device [P] bytes. */
int pvamd_cached_outside(const pvamd_grid_t* grid, const float* points, int64_t P, float level,
                         uint8_t* out, void* stream);

/* The same three queries for FLOAT64 query points.  The reference returns the query dtype (sdf.py:545-547) and torch
 * promotion makes every operation on the points float64: (points - min) / resolution and the range test (sdf.py:537,
 * 540) whatever dtype the range had (dmin / dmax / dres above), the BOUNDING_BOX branch on self.bb.to(float64)
 * (sdf.py:556-571, dbb_min / dbb_max).  Cached values are float32 and are widened exactly.
 * points: device [P][3] float64.  out_val: device [P] float64.  out_grad: device [P][3] float64.                 */
int pvamd_cached_query_f64(const pvamd_grid_t* grid, const double* points, int64_t P,
                           double* out_val, double* out_grad, uint8_t* out_oob, void* stream);
int pvamd_cached_outside_f64(const pvamd_grid_t* grid, const double* points, int64_t P, double level,
                             uint8_t* out, void* stream);
int pvamd_voxel_index_f64(const pvamd_grid_t* grid, const double* points, int64_t P,
                          int64_t* out_key, int64_t* out_flat, uint8_t* out_valid, void* stream);

/* Voxel index arithmetic alone (TorchMultidimView.ensure_index_key / ravel_multi_index / get_valid_values as
 * used at sdf.py:537-540): out_key device [P][3] int64, out_flat device [P] int64, out_valid device [P] bytes.
 * Any output may be NULL.  This is the entry point the bit-exact index tests call.                     */
int pvamd_voxel_index(const pvamd_grid_t* grid, const float* points, int64_t P,
                      int64_t* out_key, int64_t* out_flat, uint8_t* out_valid, void* stream);

/* Dense voxel containers addressed by points (VoxelGrid.__getitem__ / __setitem__, voxel.py:97-103, over the value-range
 * view voxel.py:55): the same index arithmetic as above, reading or writing a caller-owned dense array in C order
 * (float32, or bytes for torch.bool grids).  grid->vox is not used.
 * gather:  out[i] = storage[flat(points[i])] where the point passes the range test, invalid_value elsewhere.
 * scatter: storage[flat(points[i])] = values[i] (or `scalar` when values is NULL) for the points that pass the range test,
 *          others are ignored.  When several points share a voxel the LAST one in input order wins, as in a sequential
 *          loop; that needs owner_scratch: device int32 [shape0*shape1*shape2] (required iff values != NULL).      */
int pvamd_voxel_gather_f32(const pvamd_grid_t* grid, const float* storage, const float* points, int64_t P,
                           float invalid_value, float* out, void* stream);
int pvamd_voxel_gather_u8(const pvamd_grid_t* grid, const uint8_t* storage, const float* points, int64_t P,
                          uint8_t invalid_value, uint8_t* out, void* stream);
int pvamd_voxel_scatter_f32(const pvamd_grid_t* grid, float* storage, const float* points, const float* values,
                            float scalar, int64_t P, int32_t* owner_scratch, void* stream);
int pvamd_voxel_scatter_u8(const pvamd_grid_t* grid, uint8_t* storage, const float* points, const uint8_t* values,
                           uint8_t scalar, int64_t P, int32_t* owner_scratch, void* stream);

/* ComposedSDF.__call__ over CachedSDF leaves (sdf.py:392-433 + 535-571 fused; also RobotSDF.__call__,
 * model_to_sdf.py:117-125): per configuration a and point p, x_s = T[s*A+a] p for every leaf s, look up leaf
 * s at x_s, rotate that gradient back with R^T, keep the first minimum over s.
 * grids: device [S] pvamd_grid_t (every oob_mode must be BOUNDING_BOX).  tf: device [S*A][4][4] obj->leaf,
 * leaf-major (model_to_sdf.py:100-113).  out_val: device [A][P].  out_grad: device [A][P][3].
 * out_leaf: device [A][P] int32 or NULL (arg-min leaf, for tests).
 * flags: 0, or PVAMD_COMPOSED_INLINE_EXACT -- a tuning hint that never changes a result: the kernels estimate a voxel
 * index in fp32 and fall back to the reference's exact division where the estimate is within its error bound (err32 of
 * pvamd_grid_finalize) of a rounding boundary.  By default flagged points are redone after the leaf loop (cheapest when
 * flags are rare and the grids are L2-resident, i.e. the kernel is instruction-bound); with the hint the exact statements
 * sit inline (cheapest for large, gather-bound grids, whose larger coordinate / resolution ratios also flag more visits).
 * The transforms must be RIGID (orthonormal 3x3, last row 0 0 0 1): the gradient is rotated back with R^T and the
 * leaf-culling bounds rely on distances being preserved.
 * Any P >= 0 and any A >= 1 (the configuration is blockIdx.x; only the points-fastest tuning order walks slabs of 65535);
 * points / out_val / out_grad need only
 * their natural 4-byte alignment -- rows of an odd P (the reference README's M = 15,251) take the same kernel.      */
#define PVAMD_COMPOSED_INLINE_EXACT 1
#define PVAMD_COMPOSED_FORCE_PER_LANE 2   /* testing / tuning: take the one-point-per-lane kernel whatever the size */
#define PVAMD_COMPOSED_FORCE_WAVE_TILE 4  /* testing / tuning: take the wave-tile kernel whatever the size                 */
#define PVAMD_COMPOSED_POINTS_FASTEST 8   /* tuning: per-lane kernel with blocks ordered points-fastest (default: configuration-fastest) */
#define PVAMD_COMPOSED_NO_GROUPING 64   /* testing / tuning: never the chunk-grouped kernel (the round-5 kernels whatever the size) */
#define PVAMD_COMPOSED_FORCE_FUSED 128  /* testing / tuning: the chunk-grouped kernel with the in-workgroup sort whenever P >= one chunk */
#define PVAMD_COMPOSED_OUT_PACKED 32  /* pvamd_composed_query_grouped only: out_val takes [A][P] (val, gx, gy, gz) records (16-byte aligned), out_grad NULL */
#define PVAMD_COMPOSED_LEGACY_LEAF_LOOP 16 /* testing / tuning: wave-tile kernel with the round-3 leaf loop (lookups and exact roots inside the leaf loop) */
int pvamd_composed_query(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A,
                         const float* points, int64_t P,
                         float* out_val, float* out_grad, int32_t* out_leaf, int32_t flags, void* stream);

/* The same query for FLOAT64 points with a float64 transform stack (sdf.py:392-433 when the chain / the Transform3d is
 * float64: the transform, every leaf's index arithmetic, range test and bounding-box branch -- sdf.py:545-547 -- and the
 * gradient rotation are float64; cached records are widened exactly).  tf: device [S*A][4][4] float64.  points: device
 * [P][3] float64.  out_val: device [A][P] float64.  out_grad: device [A][P][3] float64.  8-byte alignment.            */
int pvamd_composed_query_f64(const pvamd_grid_t* grids, int32_t S, const double* tf, int32_t A,
                             const double* points, int64_t P,
                             double* out_val, double* out_grad, int32_t* out_leaf, void* stream);

/* The glue of ComposedSDF.__call__ for leaves that are not cached grids (MeshSDF, SphereSDF, nested compositions;
 * sdf.py:392-433 around per-leaf queries), with the fused kernel's rounding so that both paths state the same numbers:
 * pvamd_transform_points: out[a][p] = tf[a] p (sdf.py:399).  tf: device [A][4][4] -- the A transforms of ONE leaf.
 *   points: device [P][3].  out: device [A][P][3].  A <= 65535.
 * pvamd_compose_merge: fold leaf s into the running first minimum (sdf.py:409,421): where leaf_val is smaller than
 *   best_val (or is NaN against a number; everywhere when first != 0) take it, with the gradient brought back to the
 *   object frame as L^T g (L = linear part of tf[a]; valid for any affine transform).  leaf_val / best_val: device [A][P].
 *   leaf_grad / best_grad: device [A][P][3].  best_leaf: device [A][P] int32 or NULL.                                  */
int pvamd_transform_points(const float* tf, int32_t A, const float* points, int64_t P, float* out, void* stream);
int pvamd_compose_merge(const float* tf, int32_t A, int64_t P, const float* leaf_val, const float* leaf_grad,
                        int32_t s, int32_t first, float* best_val, float* best_grad, int32_t* best_leaf, void* stream);

/* The same query when the caller has sorted the points spatially (e.g. along the Morton curve of pvamd_morton_keys):
 * coherent wave tiles let whole leaves be skipped and keep the leaf-grid region a tile touches in L2, which is what
 * decides the time once the grids are far larger than L2 (README-size link grids: 4.9 -> 1.6 ms for 200 x 262,144).
 * The kernel leaves one packed (val, gx, gy, gz) record per (configuration, sorted position) in `scratch` and a second
 * pass writes out_val / out_grad in the CALLER's point order: out[a][j] = record[a][inv[j]].  Same results, bit for
 * bit, as pvamd_composed_query on the unsorted points.
 * sorted_points: device [Pp][3], the P points in processing order followed by Pp - P copies of any of them (Pp a
 * multiple of 256, 16-byte aligned).  inv: device [P] int32, position of caller point j in sorted_points.
 * scratch: device, A * Pp * 16 bytes, 16-byte aligned.  out_val: device [A][P].  out_grad: device [A][P][3].       */
int pvamd_composed_query_bucketed(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A,
                                  const float* sorted_points, const int32_t* inv, int64_t P, int64_t Pp,
                                  float* scratch, float* out_val, float* out_grad, int32_t flags, void* stream);

/* The same query with the points regrouped spatially INSIDE chunks of pvamd_group_chunk_points() consecutive points (round
 * 6; replaces nothing in the reference -- a processing order).  On scattered query points some of a wave's 64 lanes fall
 * inside a leaf's range at nearly every visit and the wave pays the look-up half for a handful of lanes; with neighbouring
 * points per wave most visits are all-outside (C4, 200 x 262,144 random points: 0.68 -> 0.5 ms).  A workgroup owns one chunk
 * and restores the caller's order through LDS, so -- unlike the globally sorted path above -- there is no second pass over
 * the outputs.  Same results, bit for bit, as pvamd_composed_query.
 * pvamd_group_points: sort every chunk of `points` (device [P][3], P >= pvamd_group_chunk_points()) once; the A
 *   configurations of a call -- and later calls on the same points -- share it.  scratch: device,
 *   pvamd_group_scratch_bytes(P) bytes, 16-byte aligned (sorted copy | bounding sphere per run of 256 | uint16 positions).
 * pvamd_composed_query_grouped: the query over that scratch.  flags: 0, or PVAMD_COMPOSED_OUT_PACKED (the records of
 *   pvamd_composed_query_packed, for any P: out_val = [A][P][4], out_grad = NULL); the INLINE_EXACT / LEGACY hints:
 *   PVAMD_E_MODE -- gather-bound grids gain nothing from it.  Any A >= 1, any P >= pvamd_group_chunk_points(), any 4-byte
 *   aligned outputs. */
int64_t pvamd_group_chunk_points(void);
int64_t pvamd_group_scratch_bytes(int64_t P);
int pvamd_group_points(const float* points, int64_t P, void* scratch, void* stream);
int pvamd_composed_query_grouped(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A, const void* scratch,
                                 int64_t P, float* out_val, float* out_grad, int32_t* out_leaf, int32_t flags, void* stream);

/* The two halves of the above, for callers that move the packed records themselves: RobotSDF.__call__
 * (model_to_sdf.py:117-125 -> sdf.py:392-433) sharded over GPUs runs on each rank's slice of the points, gathers ONE
 * buffer of records instead of val and grad, then unpacks straight into the reference's (A, P) / (A, P, 3) layout:
 * pvamd_composed_query_packed: out_rec[a][k] = (val, gx, gy, gz) of configuration a at points[k].  points: device [Pp][3],
 *   Pp a multiple of 256, at most 2^24 * 4.  out_rec: device, A * Pp * 16 bytes, 16-byte aligned.  Any A.
 * pvamd_unpack_records: out_val[a][j] / out_grad[a][j] = rec[a * stride + index[j]] for j < P (index[j] may point
 *   anywhere in the buffer, e.g. into another rank's slab).  rec: device float4 records.  index: device [P] int32.   */
int pvamd_composed_query_packed(const pvamd_grid_t* grids, int32_t S, const float* tf, int32_t A, const float* points,
                                int64_t Pp, float* out_rec, int32_t flags, void* stream);
int pvamd_unpack_records(const float* rec, const int32_t* index, int64_t P, int64_t stride, int32_t A, float* out_val,
                         float* out_grad, void* stream);

/* Prepare a mesh for the query kernels: per-triangle records (corners, original face id, bounding sphere, and the
 * triangle's in-plane bounding rectangle: centre, two unit axes, half extents) plus one bounding sphere per run of
 * PVAMD_TRI_GROUP and of PVAMD_TRI_TILE records.  The bounds only ever SKIP work that provably cannot change a result
 * (computed in float64 for the rounded values stored, inflated by abs_margin and a relative 1e-5), so query results
 * are bit-identical to the untiled brute force whatever order the triangles are given in; input in compact runs of
 * PVAMD_TRI_GROUP / PVAMD_TRI_TILE triangles (the Python host: median-split patches of the centroids, mesh_io.patch_order;
 * half the radii of Z-order runs on a surface) is what makes the skipping effective.
 * tri: device [F][3][3] fp32 soup in the order to process.  face_id: device [F] int32 original ids, or NULL for 0..F-1.
 * abs_margin: absolute slack, >= 1e-6 * (largest |vertex coordinate| + bounding-box diagonal).
 * rec_out: device float[PVAMD_REC_FLOATS(F)] (16-byte aligned).  tiles_out: device float[PVAMD_TILES_FLOATS(F)]
 * (16-byte aligned).  F <= 2^26.
 * rec_of_face_out: device [F] int32.                                                                            */
int pvamd_mesh_prepare(const float* tri, const int32_t* face_id, int32_t F, float abs_margin, float* rec_out,
                       float* tiles_out, int32_t* rec_of_face_out, void* stream);

/* Axis-aligned bounds of the finite coordinates of a point set (the box pvamd_morton_keys wants).  points: device
 * [P][3].  box_out: device [2][3] fp32 (lo xyz, hi xyz); (+inf, -inf) for a dimension without finite values.     */
int pvamd_points_aabb(const float* points, int64_t P, float* box_out, void* stream);

/* 30-bit Z-order (Morton) key of every point inside the box [lo, hi]: the sort key for a spatially coherent
 * processing order (`order` below).  points: device [P][3].  box: device [2][3] fp32 (lo xyz, hi xyz).
 * keys_out: device [P] int32.                                                                                  */
int pvamd_morton_keys(const float* points, int64_t P, const float* box, int32_t* keys_out, void* stream);

/* A spatial processing order for a point set, in one call (bounds, cell counts, scan, scatter: seven small launches
 * instead of a general-purpose device sort): order_out[k] = index of the k-th point along a HILBERT curve (since ABI 9; a
 * Z-order curve before -- the name stayed) of 16^3 cells over the points' bounding box up to 16 k points (one launch),
 * 32^3 / 64^3 / 128^3 beyond 16 k / 64 k / 1 M: consecutive cells are face neighbours at every level, so the 64 points of
 * a wave are 25 % closer together than along the Z curve.  Points of one cell come out in an arbitrary, run-dependent
 * order -- the kernels that take an `order` return the same bits for any order.
 * order_out: device [P] int32.  inv_out: device [P] int32 or NULL, inv[order[k]] = k.  sorted_points_out: device [P][3] or
 * NULL, the points in that order.  scratch: device, PVAMD_MORTON_ORDER_SCRATCH_BYTES(P) bytes, 4-byte aligned.          */
#define PVAMD_MORTON_ORDER_BITS(P) ((P) >= (1 << 20) ? 21 : ((P) >= (1 << 16) ? 18 : 15))
/* from 786,432 points on: (cell, index) pairs through a stable LSD radix sort (three 7-bit passes over 4096-pair tiles;
 * csrc/sort.hip) -- the points of a cell come out in index order.  Scratch: bounds codes, supergroup totals and the tile
 * histogram table (<= 512 words per tile), two key and two index arrays.                                             */
#ifndef PVAMD_ORDER_RADIX_SORT_FROM
#define PVAMD_ORDER_RADIX_SORT_FROM (3 << 18)  /* 786,432: counting sort 0.078 ms at 512 k, 0.166 at 1 M; radix 0.097 / 0.105 (profiles/r05_sort.txt) */
#endif
/* enough for EITHER sort, so that the size does not depend on the threshold a library was built with */
#define PVAMD_MORTON_RADIX_SCRATCH_BYTES(P) (4 * (8 + 4 * (int64_t)(P) + 512 * (((int64_t)(P) + 4095) / 4096)))
#define PVAMD_MORTON_COUNTING_SCRATCH_BYTES(P) (4 * (8 + (1 << PVAMD_MORTON_ORDER_BITS(P)) + (int64_t)(P) + 2048))
#define PVAMD_MORTON_ORDER_SCRATCH_BYTES(P) (PVAMD_MORTON_RADIX_SCRATCH_BYTES(P) > PVAMD_MORTON_COUNTING_SCRATCH_BYTES(P) \
    ? PVAMD_MORTON_RADIX_SCRATCH_BYTES(P) : PVAMD_MORTON_COUNTING_SCRATCH_BYTES(P))
int pvamd_morton_order(const float* points, int64_t P, int32_t* order_out, int32_t* inv_out, float* sorted_points_out,
                       void* scratch, void* stream);

/* ObjectFactory._do_object_frame_closest_point (sdf.py:122-172): closest surface point, signed distance by
 * ray-hit parity, gradient, face id, optional face normal.
 * mesh: host struct with device pointers.  order: device [P] int32 permutation or NULL -- the k-th lane processes
 * point order[k] (spatially sorted processing makes the tile culling effective; outputs stay in point order).  jitter_seed: counter-based replacement for the reference's unseeded
 * np.random.randn (sdf.py:149); index_base = global index of points[0], so that a query sharded across GPUs
 * draws the jitter an unsharded one would.  out_closest: device [P][3] or NULL.  out_dist: device [P].  out_grad: device
 * [P][3].  out_face: device [P] int32 or NULL.  out_normal: device [P][3] or NULL (compute_normal=True).
 * scratch: device, PVAMD_MESH_SCRATCH_BYTES(P) bytes, 8-byte aligned, or NULL.  With it, the tiles of a 64-point group
 * are spread over several workgroups where one group's serial walk would set the time -- every group of a query of up to
 * PVAMD_MESH_SCRATCH_GROUPS groups (two launches: list, then parts + outputs), and the heavy groups of a larger one
 * (points about equidistant to much of a mesh of >= 128 tiles; three launches) -- meeting in scratch (per slot: the
 * group's points, rays, bounds, best (d^2, face) and hit counts); results are the same bits either way.  Contents on
 * return are unspecified.                                                                                          */
#define PVAMD_MESH_SCRATCH_GROUPS 8192  /* point groups (of 64) a scratch buffer has slots for: every group up to this many, ... */
/* ... then 8192 or an eighth of the groups, whichever is more (C5: 2.3 % of the groups are handed over; one that finds the
 * list full walks the mesh on its own two waves, 4x slower than the rest put together when that happens to thousands) */
#define PVAMD_MESH_SCRATCH_SLOTS(P) ((((P) + 63) / 64) < PVAMD_MESH_SCRATCH_GROUPS ? (((P) + 63) / 64) : \
                                     ((((P) + 63) / 64) / 8 > PVAMD_MESH_SCRATCH_GROUPS ? (((P) + 63) / 64) / 8 : PVAMD_MESH_SCRATCH_GROUPS))
#define PVAMD_MESH_SMALL_POINTS 16384  /* pvamd_mesh_query_unordered: at most this many points */
#define PVAMD_MESH_SCRATCH_BYTES(P) (64 + PVAMD_MESH_SCRATCH_SLOTS(P) * (int64_t)(64 * 40 + 8 + 64) + 24 * PVAMD_MESH_SMALL_POINTS)
int pvamd_mesh_query(const pvamd_mesh_t* mesh, const float* points, const int32_t* order, int64_t P,
                     uint64_t jitter_seed, int64_t index_base, float* out_closest, float* out_dist, float* out_grad, int32_t* out_face,
                     float* out_normal, void* scratch, void* stream);

/* CachedSDF construction from a mesh on the device (sdf.py:498-516: the ground-truth SDF over every voxel centre of the grid
 * voxel.py:20-25 builds, then the two arrays the reference keeps), in at most three launches for a small grid: the voxel
 * centres (cartesian product of the three coordinate arrays, x slowest) and a processing order that walks them in 4 x 4 x 4
 * bricks (inside 16^3 super-bricks) are written by one kernel -- no sort --, then pvamd_mesh_query's launches, whose last one writes the packed
 * (val, gx, gy, gz) record of voxel i = (x * ny + y) * nz + z straight into the cache.  Same bits as pvamd_mesh_query over the
 * same centres followed by pvamd_pack_grid (results do not depend on the processing order; the sign jitter is indexed by i).
 * cx / cy / cz: device [nx] / [ny] / [nz] float32.  out_packed: device [nx*ny*nz][4], 16-byte aligned.
 * points_scratch: device [nx*ny*nz][3] float32.  order_scratch: device [nx*ny*nz] int32.
 * scratch: device, PVAMD_MESH_SCRATCH_BYTES(nx*ny*nz) bytes, 8-byte aligned, or NULL.                                    */
int pvamd_cache_build(const pvamd_mesh_t* mesh, const float* cx, const float* cy, const float* cz, int32_t nx, int32_t ny,
                      int32_t nz, uint64_t jitter_seed, float* out_packed, float* points_scratch, int32_t* order_scratch,
                      void* scratch, void* stream);

/* The same query for at most PVAMD_MESH_SMALL_POINTS points that come without a processing order: the order is worked out
 * inside (into order_scratch: device [P] int32, contents on return = the order used), by one workgroup of a launch whose
 * other workgroups already do the per-point work that does not need it -- a separate pvamd_morton_order call in front of
 * pvamd_mesh_query costs a 10,000-point query a fifth of its time.  Same results, bit for bit.                       */
int pvamd_mesh_query_unordered(const pvamd_mesh_t* mesh, const float* points, int64_t P, uint64_t jitter_seed,
                               int64_t index_base, float* out_closest, float* out_dist, float* out_grad, int32_t* out_face,
                               float* out_normal, int32_t* order_scratch, void* scratch, void* stream);

/* The draw of sample_mesh_points (sdf.py:643-650: open3d's sample_points_uniformly + a random subset), counter-based so
 * that it is reproducible: sample i picks the triangle t with cdf[t-1] <= u < cdf[t] (u, r1, r2 = 53-bit uniforms from
 * splitmix64(seed, i)) and the point (1 - sqrt(r1)) a + sqrt(r1)(1 - r2) b + sqrt(r1) r2 c, in float64.
 * tri: device [F][3][3] fp32 corners (any order).  cdf: device [F] float64 inclusive cumulative area fractions
 * (non-decreasing, cdf[F-1] >= 1).  out_points: device [n][3] float64.  out_face: device [n] int32 index into tri, or
 * NULL.  out_key: device [n] int64 non-negative random keys (the n' smallest select a uniform random subset), or NULL. */
int pvamd_sample_surface(const float* tri, const double* cdf, int32_t F, int64_t n, uint64_t seed,
                         double* out_points, int32_t* out_face, int64_t* out_key, void* stream);

/* batch_chamfer_dist (chamfer.py:79-94) against a mesh: for each of B world->object transforms, transform the
 * N points, unsigned distance to the mesh, accumulate sum_n (scale*d)^2.  The caller divides by the GLOBAL N
 * (after an all-reduce when the points are sharded across GPUs).
 * W: device [B][4][4].  points: device [N][3].  order: as for pvamd_mesh_query, or NULL.
 * out_sum: device [B] float64, ZEROED by this call.  scratch: device, PVAMD_MESH_SCRATCH_BYTES(N) bytes, 8-byte aligned,
 * or NULL (as for pvamd_mesh_query: heavy point groups are spread over several workgroups; same sums up to the order of
 * the float64 additions).                                                                                     */
int pvamd_chamfer_mesh(const pvamd_mesh_t* mesh, const float* W, int32_t B, const float* points,
                       const int32_t* order, int64_t N, float scale, double* out_sum, void* scratch, void* stream);

/* The same sums when the caller has ALREADY transformed the points: points holds x[b][n] = W[b] p[n] for all B transforms
 * ([B][per][3], e.g. from pvamd_transform_points, whose rounding is the chamfer kernel's), and `order` walks all B * per
 * of them in ONE spatial order (pvamd_morton_order over the flat array).  Point i adds to out_sum[i / per].  What
 * pairwise_distance_chamfer / PlausibleDiversity want (chamfer.py:20-59,173-183: 10^4 transforms x 500 model points): with
 * few points per transform the 64 points a wave of pvamd_chamfer_mesh shares are a whole patch of the object apart and its
 * bounds cull little; in one global order they are neighbours (100 x 100 poses x 500 points on the drill: 12.3 -> 3.5 ms).
 * out_sum: device [B] float64, zeroed by this call.  scratch: PVAMD_MESH_SCRATCH_BYTES(B * per) bytes or NULL.          */
int pvamd_chamfer_mesh_flat(const pvamd_mesh_t* mesh, int32_t B, const float* points, const int32_t* order, int64_t per,
                            float scale, double* out_sum, void* scratch, void* stream);

/* Same against a cached grid (obj_sdf branch, chamfer.py:84-85).  grid: host.                            */
int pvamd_chamfer_grid(const pvamd_grid_t* grid, const float* W, int32_t B, const float* points, int64_t N,
                       float scale, double* out_sum, void* stream);

/* RobotSDF.set_joint_configuration's contraction (model_to_sdf.py:104-113): out[s*A+a] = offset_inv[s] @
 * rigid_inverse(link_world[s*A+a]).  offset_inv: device [S][4][4].  link_world: device [S*A][4][4] leaf-major.
 * out: device [S*A][4][4].  The 4x4x4 products run on the f32 MFMA (v_mfma_f32_4x4x1_16b_f32).            */
int pvamd_transform_stack(const float* offset_inv, const float* link_world, int32_t S, int32_t A,
                          float* out, void* stream);

/* Forward kinematics for A configurations (RobotSDF.set_joint_configuration, model_to_sdf.py:94-102):
 * world[f] = world[parent f] @ origin[f] @ motion(joint f, q).  joints: device [F].  q, sin_q, cos_q: device [A][M]
 * (joint values and their sine / cosine).  scratch: device [F][12][A] fp32 (all frames' matrices, rows 0..2, kept for
 * children to read).  link_world_out: device [S*A][4][4], leaf-major, the input of pvamd_transform_stack.          */
int pvamd_chain_fk(const pvamd_joint_t* joints, int32_t F, const float* q, const float* sin_q, const float* cos_q,
                   int32_t A, int32_t M, float* scratch, float* link_world_out, void* stream);

/* PlausibleDiversity's reduction of the (B, P) pairwise chamfer matrix (chamfer.py:185-195) in one pass: per row / per column
 * the minimum and where it is (first index on ties, a NaN counts as the minimum: torch.min), and means[0] = mean of the row
 * minima (plausibility), means[1] = mean of the column minima (coverage), summed in float64 in a fixed order.
 * errors: device [B][P] float32 (is_f64 = 0) or float64.  row_val / col_val: device [B] / [P] of the same dtype.
 * row_idx / col_idx: device int64.  means: device [2] float64.                                                          */
int pvamd_pairwise_min_reduce(const void* errors, int32_t is_f64, int32_t B, int32_t P, void* row_val, int64_t* row_idx,
                              void* col_val, int64_t* col_idx, double* means, void* stream);

/* The whole of RobotSDF.set_joint_configuration (model_to_sdf.py:94-113) in ONE launch, from joint values that already sit
 * on the device: sin / cos, the frame walk of pvamd_chain_fk, and stack_out[s*A+a] = offset_inv[s] @
 * rigid_inverse(world[leaf s, a]) (the f32-MFMA statement of pvamd_transform_stack) -- the obj->leaf stack that
 * pvamd_composed_query consumes.  q: device [A][M].  offset_inv: device [S][4][4].  sincos_out: device [A][M][2] (sin, cos
 * of every revolute joint value, as used: feed them to a CPU restatement to reproduce the stack bit for bit) or NULL.
 * scratch: device [F][12][A] fp32.  link_world_out: device [S*A][4][4] leaf-major or NULL.  stack_out: device [S*A][4][4].
 * At most 50 SDF-carrying links (the leaf frames of 64 configurations are staged in LDS).  Graph-capturable: no host
 * data, no allocation, one kernel.                                                                                   */
int pvamd_configure_chain(const pvamd_joint_t* joints, int32_t F, const float* q, int32_t A, int32_t M,
                          const float* offset_inv, int32_t S, float* sincos_out, float* scratch,
                          float* link_world_out, float* stack_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PVAMD_H */
