/* CPU oracle -- TEST INFRASTRUCTURE, not part of the product (see pvamd_oracle.c).  The grid description and the entry points a
 * plain-C checker links against (tests/cabi/cabi_check.c); the Python binding (oracle/oracle.py) mirrors the same layout. */
#ifndef PVAMD_ORACLE_H
#define PVAMD_ORACLE_H
#include <stdint.h>

/* ------------------------------------------------------------------------------------------------------------
 * Grid description used by the oracle: the reference's OWN layout (separate val [nx,ny,nz] and grad [n,3]
 * arrays, sdf.py:504-505,521-523), not the packed layout of the product.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct oracle_grid {
    const float* val;   /* [nx*ny*nz]      */
    const float* grad;  /* [nx*ny*nz][3]   */
    double dmin[3], dmax[3], dres[3];
    float fmin[3], fmax[3], fres[3];
    float bb_min[3], bb_max[3];
    int32_t shape[3];
    int32_t index_f64;
    int32_t oob_mode; /* 0 LOOKUP_GT_SDF (zeros + mask), 1 BOUNDING_BOX */
    int32_t rule;     /* which of the UNPINNED choices of the third-party view to restate (include/pvamd.h PVAMD_RULE_*):
                         0 = round half to even + validity on the value; 1 validity on the rounded index; 2 round half away
                         from zero; 4 floor(q + 0.5); 8 (host side only) resolution of a float32 range evaluated in float64 */
    double dbb_min[3], dbb_max[3]; /* the bounding box as float64: what sdf.py:556-557 casts self.bb to for float64 queries */
} oracle_grid_t;

/* A triangle mesh as the reference holds it after precompute_sdf (sdf.py:97-120): soup + per-face unit normals. */
typedef struct oracle_mesh {
    const float* tri;    /* [F][3][3] */
    const float* normal; /* [F][3]    */
    int32_t F;
    int32_t reserved;
    double ray_dir[3];   /* bounding_box(padding=1.0)[:,1], sdf.py:147 */
} oracle_mesh_t;

void oracle_cached_query(const oracle_grid_t* g, const float* pts, int64_t P, float* out_val, float* out_grad, uint8_t* out_oob);
void oracle_voxel_index(const oracle_grid_t* g, const float* pts, int64_t P, int64_t* out_key, int64_t* out_flat, uint8_t* out_valid);
void oracle_composed_query(const oracle_grid_t* grids, int32_t S, const float* tf, int32_t A, const float* pts, int64_t P,
                           float* out_val, float* out_grad, int32_t* out_leaf);
void oracle_mesh_query(const oracle_mesh_t* m, const float* pts, int64_t P, uint64_t seed, int64_t index_base, float* out_closest,
                       float* out_dist, float* out_grad, int32_t* out_face, float* out_normal);
void oracle_chamfer_mesh(const oracle_mesh_t* m, const float* W, int32_t B, const float* pts, int64_t N, float scale, double* out_sum);
void oracle_chamfer_grid(const oracle_grid_t* g, const float* W, int32_t B, const float* pts, int64_t N, float scale, double* out_sum);
void oracle_transform_stack(const float* offset_inv, const float* link_world, int32_t S, int32_t A, float* out);
#endif
