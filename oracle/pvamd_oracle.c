/*
 * pvamd_oracle.c -- CPU restatement of the reference's SDF-query hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  Nothing under pytorch_volumetric_amd/ may import, link or call it.
 *
 * PARITY STATUS: "parity unpinned" for the third-party arithmetic.  The reference (UM-ARM-Lab/pytorch_volumetric
 * 0.5.2) cannot be imported in the build container (open3d, multidim_indexing, pytorch_kinematics,
 * arm_pytorch_utilities are absent, un-vendored and un-pinned -- pyproject.toml:55-60), and it ships no golden
 * vectors.  This file restates (a) the reference's own glue, line by line, and (b) the published algorithms of
 * the absent dependencies at the reference's call sites:
 *   - multidim_indexing TorchMultidimView (no version pin): value->index = round_half_even((p-min)/res),
 *     res=(max-min)/(shape-1), validity = min<=p<=max inclusive, C-order ravel      [sdf.py:521,537-540]
 *   - open3d RaycastingScene (no version pin) -> Embree: closest point = Ericson, "Real-Time Collision
 *     Detection" 5.1.5 as in Embree's tutorials/common/math/closest_point.h; ray hit = Embree's Moeller-Trumbore
 *     triangle intersector, rays are [origin, DIRECTION], tnear=0, tfar=inf         [sdf.py:134,153]
 *   - pytorch_kinematics >=0.5.6 Transform3d: column-vector 4x4, transform_points, transform_normals
 *                                                                                   [sdf.py:399,409; chamfer.py:81-82]
 * It is pinned (tests/test_oracle_*.py) against: closed-form SDFs (sphere, cube [-1,1]^3 = the reference's
 * tests/pv_sdf_debug/box_template.obj), the property assertions of the reference's own tests, and golden vectors
 * produced by running the liftable parts of the reference in the build container (tests/golden/make_golden.py).
 *
 * Floating point: every fused multiply-add below is explicit (fmaf/fma); build with -ffp-contract=off so the
 * compiler adds none.  The HIP kernels state the same operation sequences, so GPU-vs-oracle differences are
 * expected to be exactly zero except where noted in DESIGN.md.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "pvamd_oracle.h" /* oracle_grid_t: the reference's OWN layout (separate val and grad arrays), and the prototypes a C checker uses */

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* The quotient -> index step.  Which rounding multidim_indexing applies is not pinned by anything in the reference
 * (sdf.py:537 only calls ensure_index_key); rule 0 restates torch.round (half to even), the bits the alternatives. */
static double round_rule_d(int rule, double q) {
    if (rule & 2) return round(q);        /* halves away from zero */
    if (rule & 4) return floor(q + 0.5);  /* the sum rounds in float64 */
    return rint(q);
}
static float round_rule_f(int rule, float q) {
    if (rule & 2) return roundf(q);
    if (rule & 4) return floorf(q + 0.5f); /* the sum rounds in float32 (-ffp-contract=off) */
    return rintf(q);
}
/* float -> int64 as the GPU converts (saturating, NaN -> 0), so that keys of invalid points compare equal too */
static int64_t to_key(double kq) {
    if (!(kq == kq)) return 0;
    if (kq >= 9223372036854775807.0) return INT64_MAX;
    if (kq <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)kq;
}

/* sdf.py:537 ensure_index_key + sdf.py:540 get_valid_values, one dimension */
static int index_1d(const oracle_grid_t* g, int d, float p, int64_t* k) {
    if (g->index_f64) {
        const double pd = (double)p;
        const double kq = round_rule_d(g->rule, (pd - g->dmin[d]) / g->dres[d]);
        *k = to_key(kq);
        if (g->rule & 1) return (kq >= 0.0) && (kq <= (double)(g->shape[d] - 1)); /* validity on the rounded index */
        return (g->dmin[d] <= pd) && (pd <= g->dmax[d]);
    }
    const float kq = round_rule_f(g->rule, (p - g->fmin[d]) / g->fres[d]);
    *k = to_key((double)kq);
    if (g->rule & 1) return (kq >= 0.f) && (kq <= (float)(g->shape[d] - 1));
    return (g->fmin[d] <= p) && (p <= g->fmax[d]);
}

void oracle_voxel_index(const oracle_grid_t* g, const float* pts, int64_t P, int64_t* out_key, int64_t* out_flat,
                        uint8_t* out_valid) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        int64_t k[3];
        int valid = 1;
        for (int d = 0; d < 3; ++d) valid &= index_1d(g, d, pts[3 * i + d], &k[d]);
        if (out_key) {
            out_key[3 * i] = k[0];
            out_key[3 * i + 1] = k[1];
            out_key[3 * i + 2] = k[2];
        }
        /* sdf.py:538 ravel_multi_index, C order */
        if (out_flat) out_flat[i] = (k[0] * g->shape[1] + k[1]) * g->shape[2] + k[2];
        if (out_valid) out_valid[i] = (uint8_t)valid;
    }
}

/* sdf.py:559-571 BOUNDING_BOX fallback for one point */
static void bounding_box_sdf(const oracle_grid_t* g, const float* p, float* val, float* grad) {
    float t[3];
    for (int d = 0; d < 3; ++d) {
        float dmin = g->bb_min[d] - p[d]; /* :559 */
        const int dmin_active = dmin > 0.f; /* :560 */
        if (!dmin_active) dmin = 0.f;       /* :561 */
        float dmax = p[d] - g->bb_max[d];   /* :562 */
        if (!(dmax > 0.f)) dmax = 0.f;      /* :563-564 */
        float dtotal = dmin + dmax;         /* :565 */
        if (dmin_active) dtotal = -dtotal;  /* :567 */
        t[d] = dtotal;
    }
    const float n = sqrtf(fmaf(t[2], t[2], fmaf(t[1], t[1], t[0] * t[0]))); /* :568 */
    grad[0] = t[0] / n; /* :570 (0/0 -> NaN like the reference) */
    grad[1] = t[1] / n;
    grad[2] = t[2] / n;
    *val = n; /* :571 */
}

/* one CachedSDF lookup; returns validity */
static int cached_lookup(const oracle_grid_t* g, const float* p, float* val, float* grad) {
    int64_t k[3];
    int valid = 1;
    for (int d = 0; d < 3; ++d) valid &= index_1d(g, d, p[d], &k[d]);
    if (valid) {
        const int64_t flat = (k[0] * g->shape[1] + k[1]) * g->shape[2] + k[2];
        *val = g->val[flat];           /* :549 */
        grad[0] = g->grad[3 * flat];   /* :550 */
        grad[1] = g->grad[3 * flat + 1];
        grad[2] = g->grad[3 * flat + 2];
    } else if (g->oob_mode == 1) {
        bounding_box_sdf(g, p, val, grad);
    } else {
        *val = 0.f; /* :546-547 zeros; the caller queries gt_sdf on this subset (:553-554) */
        grad[0] = grad[1] = grad[2] = 0.f;
    }
    return valid;
}

/* CachedSDF.__call__, sdf.py:535-571 */
void oracle_cached_query(const oracle_grid_t* g, const float* pts, int64_t P, float* out_val, float* out_grad,
                         uint8_t* out_oob) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        const int valid = cached_lookup(g, pts + 3 * i, out_val + i, out_grad + 3 * i);
        if (out_oob) out_oob[i] = (uint8_t)!valid;
    }
}

/* ---- float64 query points ----
 * The reference's output dtype is the query dtype (sdf.py:545-547) and every tensor op on the points promotes to it:
 * (points - min) / resolution (sdf.py:537 via the view) is float64 whatever dtype the range had, the range test
 * (sdf.py:540) compares in float64, and the BOUNDING_BOX branch runs on self.bb.to(float64) (sdf.py:556-557). */
static int index_1d_f64(const oracle_grid_t* g, int d, double p, int64_t* k) {
    const double kq = round_rule_d(g->rule, (p - g->dmin[d]) / g->dres[d]); /* dmin / dres = the view's tensors promoted to float64 */
    *k = to_key(kq);
    if (g->rule & 1) return (kq >= 0.0) && (kq <= (double)(g->shape[d] - 1));
    return (g->dmin[d] <= p) && (p <= g->dmax[d]);
}

void oracle_voxel_index_f64(const oracle_grid_t* g, const double* pts, int64_t P, int64_t* out_key, int64_t* out_flat,
                            uint8_t* out_valid) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        int64_t k[3];
        int valid = 1;
        for (int d = 0; d < 3; ++d) valid &= index_1d_f64(g, d, pts[3 * i + d], &k[d]);
        if (out_key) {
            out_key[3 * i] = k[0];
            out_key[3 * i + 1] = k[1];
            out_key[3 * i + 2] = k[2];
        }
        if (out_flat) out_flat[i] = (k[0] * g->shape[1] + k[1]) * g->shape[2] + k[2];
        if (out_valid) out_valid[i] = (uint8_t)valid;
    }
}

/* one CachedSDF lookup for a float64 point (sdf.py:535-571 with every tensor op promoted to float64); returns validity */
static int cached_lookup_f64(const oracle_grid_t* g, const double* p, double* val, double* grad) {
    int64_t k[3];
    int valid = 1;
    for (int d = 0; d < 3; ++d) valid &= index_1d_f64(g, d, p[d], &k[d]);
    if (valid) {
        const int64_t flat = (k[0] * g->shape[1] + k[1]) * g->shape[2] + k[2];
        *val = (double)g->val[flat];           /* :549 float32 cache assigned into a float64 tensor */
        grad[0] = (double)g->grad[3 * flat];   /* :550 */
        grad[1] = (double)g->grad[3 * flat + 1];
        grad[2] = (double)g->grad[3 * flat + 2];
    } else if (g->oob_mode == 1) {
        double t[3];
        for (int d = 0; d < 3; ++d) {
            double dmin = g->dbb_min[d] - p[d];  /* :559 */
            const int dmin_active = dmin > 0.0;  /* :560 */
            if (!dmin_active) dmin = 0.0;        /* :561 */
            double dmax = p[d] - g->dbb_max[d];  /* :562 */
            if (!(dmax > 0.0)) dmax = 0.0;       /* :563-564 */
            double dtotal = dmin + dmax;         /* :565 */
            if (dmin_active) dtotal = -dtotal;   /* :567 */
            t[d] = dtotal;
        }
        const double n = sqrt(fma(t[2], t[2], fma(t[1], t[1], t[0] * t[0]))); /* :568 */
        grad[0] = t[0] / n; /* :570 */
        grad[1] = t[1] / n;
        grad[2] = t[2] / n;
        *val = n; /* :571 */
    } else {
        *val = 0.0;
        grad[0] = grad[1] = grad[2] = 0.0;
    }
    return valid;
}

void oracle_cached_query_f64(const oracle_grid_t* g, const double* pts, int64_t P, double* out_val, double* out_grad,
                             uint8_t* out_oob) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        const int valid = cached_lookup_f64(g, pts + 3 * i, out_val + i, out_grad + 3 * i);
        if (out_oob) out_oob[i] = (uint8_t)!valid;
    }
}

void oracle_cached_outside_f64(const oracle_grid_t* g, const double* pts, int64_t P, double level, uint8_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        int64_t k[3];
        int valid = 1;
        for (int d = 0; d < 3; ++d) valid &= index_1d_f64(g, d, pts[3 * i + d], &k[d]);
        if (valid) {
            const int64_t flat = (k[0] * g->shape[1] + k[1]) * g->shape[2] + k[2];
            /* sdf.py:601 raw_data[flat] > surface_level: a float32 tensor against a python scalar compares in float32 */
            out[i] = (uint8_t)(g->val[flat] > (float)level);
        } else {
            out[i] = 1;
        }
    }
}

/* CachedSDF.outside_surface, sdf.py:593-602 */
void oracle_cached_outside(const oracle_grid_t* g, const float* pts, int64_t P, float level, uint8_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        int64_t k[3];
        int valid = 1;
        for (int d = 0; d < 3; ++d) valid &= index_1d(g, d, pts[3 * i + d], &k[d]);
        if (valid) {
            const int64_t flat = (k[0] * g->shape[1] + k[1]) * g->shape[2] + k[2];
            out[i] = (uint8_t)(g->val[flat] > level);
        } else {
            out[i] = 1;
        }
    }
}

/* x = M p, row-major 4x4, column-vector convention (chamfer.py:14); k-ordered fma chain (bmm k-loop) */
static void affine_apply(const float* M, const float* p, float* x) {
    for (int r = 0; r < 3; ++r) {
        const float* m = M + 4 * r;
        x[r] = fmaf(m[2], p[2], fmaf(m[1], p[1], m[0] * p[0])) + m[3];
    }
}

/* The glue of ComposedSDF.__call__ around arbitrary leaves (sdf.py:399 transform, :409 normals back, :421 argmin) */
void oracle_transform_points(const float* tf, int32_t A, const float* pts, int64_t P, float* out) {
    for (int32_t a = 0; a < A; ++a)
        for (int64_t i = 0; i < P; ++i) affine_apply(tf + 16 * (int64_t)a, pts + 3 * i, out + 3 * ((int64_t)a * P + i));
}

void oracle_compose_merge(const float* tf, int32_t A, int64_t P, const float* leaf_val, const float* leaf_grad, int32_t s,
                          int32_t first, float* best_val, float* best_grad, int32_t* best_leaf) {
    for (int32_t a = 0; a < A; ++a) {
        const float* M = tf + 16 * (int64_t)a;
        for (int64_t i = 0; i < P; ++i) {
            const int64_t o = (int64_t)a * P + i;
            const float v = leaf_val[o];
            const int take = first || (v < best_val[o]) || (isnan(v) && !isnan(best_val[o]));
            if (!take) continue;
            const float* g = leaf_grad + 3 * o;
            best_val[o] = v;
            for (int j = 0; j < 3; ++j) best_grad[3 * o + j] = fmaf(M[8 + j], g[2], fmaf(M[4 + j], g[1], M[j] * g[0]));
            if (best_leaf) best_leaf[o] = s;
        }
    }
}

/* ComposedSDF.__call__ over CachedSDF leaves, sdf.py:392-433 (+ RobotSDF.__call__, model_to_sdf.py:117-125).
 * tf: [S*A][16] obj->leaf, leaf-major.  out_val [A][P], out_grad [A][P][3], out_leaf [A][P] or NULL. */
void oracle_composed_query(const oracle_grid_t* grids, int32_t S, const float* tf, int32_t A, const float* pts,
                           int64_t P, float* out_val, float* out_grad, int32_t* out_leaf) {
#pragma omp parallel for schedule(static) collapse(2)
    for (int32_t a = 0; a < A; ++a) {
        for (int64_t i = 0; i < P; ++i) {
            float best_v = 0.f, best_g[3] = {0.f, 0.f, 0.f};
            int32_t best_s = -1;
            for (int32_t s = 0; s < S; ++s) {
                const float* M = tf + 16 * ((int64_t)s * A + a);
                float x[3], v, g[3];
                affine_apply(M, pts + 3 * i, x);   /* :399 */
                cached_lookup(&grids[s], x, &v, g); /* :407 */
                /* torch.argmin (:421): first minimum wins; a NaN counts as the minimum */
                const int take = (best_s < 0) || (v < best_v) || (isnan(v) && !isnan(best_v));
                if (take) {
                    best_v = v;
                    best_s = s;
                    /* :409 transform_normals by leaf->obj = R^T of the (rigid) obj->leaf rotation */
                    for (int j = 0; j < 3; ++j)
                        best_g[j] = fmaf(M[8 + j], g[2], fmaf(M[4 + j], g[1], M[j] * g[0]));
                }
            }
            const int64_t o = (int64_t)a * P + i;
            out_val[o] = best_v;
            out_grad[3 * o] = best_g[0];
            out_grad[3 * o + 1] = best_g[1];
            out_grad[3 * o + 2] = best_g[2];
            if (out_leaf) out_leaf[o] = best_s;
        }
    }
}

/* The same composition for FLOAT64 query points and a float64 transform stack (a RobotSDF over a float64 chain, or a
 * float64 Transform3d handed to ComposedSDF): sdf.py:399 transforms in float64, every leaf answers in the query dtype
 * (sdf.py:545-547: index arithmetic, range test and bounding-box branch in float64), sdf.py:409 rotates the float64
 * gradient back, sdf.py:421 takes the first minimum.  tf: [S*A][16] float64. */
void oracle_composed_query_f64(const oracle_grid_t* grids, int32_t S, const double* tf, int32_t A, const double* pts,
                               int64_t P, double* out_val, double* out_grad, int32_t* out_leaf) {
#pragma omp parallel for schedule(static) collapse(2)
    for (int32_t a = 0; a < A; ++a) {
        for (int64_t i = 0; i < P; ++i) {
            double best_v = 0.0, best_g[3] = {0.0, 0.0, 0.0};
            int32_t best_s = -1;
            const double* p = pts + 3 * i;
            for (int32_t s = 0; s < S; ++s) {
                const double* M = tf + 16 * ((int64_t)s * A + a);
                double x[3], v, g[3];
                for (int r = 0; r < 3; ++r) /* :399, the k-ordered chain of the float32 statement, in float64 */
                    x[r] = fma(M[4 * r + 2], p[2], fma(M[4 * r + 1], p[1], M[4 * r] * p[0])) + M[4 * r + 3];
                cached_lookup_f64(&grids[s], x, &v, g); /* :407 */
                const int take = (best_s < 0) || (v < best_v) || (isnan(v) && !isnan(best_v)); /* :421 */
                if (take) {
                    best_v = v;
                    best_s = s;
                    for (int j = 0; j < 3; ++j) /* :409 */
                        best_g[j] = fma(M[8 + j], g[2], fma(M[4 + j], g[1], M[j] * g[0]));
                }
            }
            const int64_t o = (int64_t)a * P + i;
            out_val[o] = best_v;
            out_grad[3 * o] = best_g[0];
            out_grad[3 * o + 1] = best_g[1];
            out_grad[3 * o + 2] = best_g[2];
            if (out_leaf) out_leaf[o] = best_s;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Mesh query: sdf.py:122-172
 * ---------------------------------------------------------------------------------------------------------- */
/* oracle_mesh_t: pvamd_oracle.h */

static float dot3(const float* u, const float* v) { return fmaf(u[2], v[2], fmaf(u[1], v[1], u[0] * v[0])); }
static void sub3(const float* u, const float* v, float* o) {
    o[0] = u[0] - v[0];
    o[1] = u[1] - v[1];
    o[2] = u[2] - v[2];
}
static void cross3(const float* u, const float* v, float* o) {
    o[0] = fmaf(u[1], v[2], -(u[2] * v[1]));
    o[1] = fmaf(u[2], v[0], -(u[0] * v[2]));
    o[2] = fmaf(u[0], v[1], -(u[1] * v[0]));
}

/* Ericson RTCD 5.1.5 / Embree tutorials closest_point.h closestPointTriangle */
static void closest_point_triangle(const float* p, const float* a, const float* b, const float* c, float* q) {
    float ab[3], ac[3], ap[3];
    sub3(b, a, ab);
    sub3(c, a, ac);
    sub3(p, a, ap);
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { memcpy(q, a, 12); return; }
    float bp[3];
    sub3(p, b, bp);
    const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { memcpy(q, b, 12); return; }
    float cp[3];
    sub3(p, c, cp);
    const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { memcpy(q, c, 12); return; }
    const float vc = fmaf(d1, d4, -(d3 * d2));
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        const float v = d1 / (d1 - d3);
        for (int k = 0; k < 3; ++k) q[k] = fmaf(v, ab[k], a[k]);
        return;
    }
    const float vb = fmaf(d5, d2, -(d1 * d6));
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        const float v = d2 / (d2 - d6);
        for (int k = 0; k < 3; ++k) q[k] = fmaf(v, ac[k], a[k]);
        return;
    }
    const float va = fmaf(d3, d6, -(d5 * d4));
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        float bc[3];
        sub3(c, b, bc);
        for (int k = 0; k < 3; ++k) q[k] = fmaf(v, bc[k], b[k]);
        return;
    }
    const float denom = 1.f / ((va + vb) + vc);
    const float v = vb * denom, w = vc * denom;
    for (int k = 0; k < 3; ++k) q[k] = fmaf(w, ac[k], fmaf(v, ab[k], a[k]));
}

/* Embree MoellerTrumboreIntersector1 (kernels/geometry/triangle_intersector_moeller.h), ray = org + t*dir,
 * tnear = 0 (strict), tfar = +inf */
static int ray_hits_triangle(const float* org, const float* dir, const float* v0, const float* v1, const float* v2) {
    float e1[3], e2[3], Ng[3], C[3], R[3];
    sub3(v0, v1, e1);
    sub3(v2, v0, e2);
    cross3(e2, e1, Ng);
    sub3(v0, org, C);
    cross3(C, dir, R);
    const float den = dot3(Ng, dir);
    if (den == 0.f) return 0;
    const float absden = fabsf(den);
    const float sgn = den < 0.f ? -1.f : 1.f;
    const float U = dot3(R, e2) * sgn;
    const float V = dot3(R, e1) * sgn;
    if (!(U >= 0.f) || !(V >= 0.f) || !(U + V <= absden)) return 0;
    const float T = dot3(Ng, C) * sgn;
    return T > 0.f; /* absden*tnear < T  and  T <= absden*inf */
}

static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Counter-based stand-in for the reference's unseeded np.random.randn (sdf.py:149): an Irwin-Hall(12) variate
 * built from twelve 16-bit uniforms, exactly representable in fp32 (unit variance, |g| <= 6). */
static float jitter_normal(uint64_t seed, int64_t index, int c) {
    int32_t s = 0;
    for (int k = 0; k < 3; ++k) {
        const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)index * 9u + (uint64_t)(c * 3 + k)));
        s += (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) +
             (int32_t)((h >> 48) & 0xFFFF);
    }
    return (float)(s - 393210) * (1.0f / 65536.0f);
}

/* sample_mesh_points' draw (sdf.py:643-650), counter-based: see pvamd_sample_surface in include/pvamd.h */
static double uniform01(uint64_t seed, int64_t index, int k) {
    const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)index * 4u + (uint64_t)k));
    return (double)(h >> 11) * 0x1.0p-53;
}

void oracle_sample_surface(const float* tri, const double* cdf, int32_t F, int64_t n, uint64_t seed, double* out_points,
                           int32_t* out_face, int64_t* out_key) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double u = uniform01(seed, i, 0);
        int32_t lo = 0, hi = F - 1;
        while (lo < hi) { /* first triangle whose cumulative area fraction exceeds u */
            const int32_t mid = (lo + hi) >> 1;
            if (cdf[mid] > u) hi = mid;
            else lo = mid + 1;
        }
        const float* t = tri + 9 * (int64_t)lo;
        const double s = sqrt(uniform01(seed, i, 1)), r2 = uniform01(seed, i, 2);
        const double wa = 1.0 - s, wb = s * (1.0 - r2), wc = s * r2;
        for (int d = 0; d < 3; ++d)
            out_points[3 * i + d] = fma(wc, (double)t[6 + d], fma(wb, (double)t[3 + d], wa * (double)t[d]));
        if (out_face) out_face[i] = lo;
        if (out_key) out_key[i] = (int64_t)(splitmix64(seed ^ splitmix64((uint64_t)i * 4u + 3u)) >> 1);
    }
}

void oracle_jitter_dir(const oracle_mesh_t* m, uint64_t seed, int64_t index, float* dir) {
    for (int c = 0; c < 3; ++c)
        dir[c] = (float)fma(1e-4, (double)jitter_normal(seed, index, c), m->ray_dir[c]); /* :147-150 */
}

/* ObjectFactory._do_object_frame_closest_point, sdf.py:122-172.  index_base: global index of pts[0] (so a
 * sharded query draws the same jitter as an unsharded one). */
void oracle_mesh_query(const oracle_mesh_t* m, const float* pts, int64_t P, uint64_t seed, int64_t index_base,
                       float* out_closest, float* out_dist, float* out_grad, int32_t* out_face, float* out_normal) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < P; ++i) {
        const float* p = pts + 3 * i;
        /* :134 compute_closest_points -- brute force; smallest squared distance, lowest face id on ties */
        float best_d2 = INFINITY, best_q[3] = {NAN, NAN, NAN};
        int32_t best_f = -1;
        float dir[3];
        oracle_jitter_dir(m, seed, index_base + i, dir);
        int hits = 0;
        for (int32_t f = 0; f < m->F; ++f) {
            const float* t = m->tri + 9 * (int64_t)f;
            float q[3], g[3];
            closest_point_triangle(p, t, t + 3, t + 6, q);
            sub3(q, p, g);
            const float d2 = dot3(g, g);
            if (d2 < best_d2) {
                best_d2 = d2;
                best_f = f;
                memcpy(best_q, q, 12);
            }
            hits += ray_hits_triangle(p, dir, t, t + 3, t + 6); /* :153 count_intersections */
        }
        float g[3];
        sub3(best_q, p, g);                                              /* :139 */
        float d = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);      /* :141 np.linalg.norm */
        if (d > 0.f) {                                                   /* :143-144 */
            g[0] /= d;
            g[1] /= d;
            g[2] /= d;
        }
        const int inside = hits & 1;                                     /* :154 */
        if (inside) d = -d;                                              /* :155 */
        else { g[0] = -g[0]; g[1] = -g[1]; g[2] = -g[2]; }               /* :157 */
        if (fabsf(d) < 1e-3f && best_f >= 0) {                           /* :162-164 */
            g[0] = m->normal[3 * best_f];
            g[1] = m->normal[3 * best_f + 1];
            g[2] = m->normal[3 * best_f + 2];
        }
        if (out_closest) memcpy(out_closest + 3 * i, best_q, 12);
        out_dist[i] = d;
        memcpy(out_grad + 3 * i, g, 12);
        if (out_face) out_face[i] = best_f;
        if (out_normal) {                                                /* :169-171 */
            for (int k = 0; k < 3; ++k) out_normal[3 * i + k] = best_f >= 0 ? m->normal[3 * best_f + k] : NAN;
        }
    }
}

/* unsigned distance only (chamfer squares it) */
static float mesh_distance(const oracle_mesh_t* m, const float* p) {
    float best_d2 = INFINITY, best_q[3] = {NAN, NAN, NAN};
    for (int32_t f = 0; f < m->F; ++f) {
        const float* t = m->tri + 9 * (int64_t)f;
        float q[3], g[3];
        closest_point_triangle(p, t, t + 3, t + 6, q);
        sub3(q, p, g);
        const float d2 = dot3(g, g);
        if (d2 < best_d2) {
            best_d2 = d2;
            memcpy(best_q, q, 12);
        }
    }
    float g[3];
    sub3(best_q, p, g);
    return sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
}

/* batch_chamfer_dist, chamfer.py:79-94, mesh branch.  out_sum[b] = sum_n (scale*d)^2 in float64. */
void oracle_chamfer_mesh(const oracle_mesh_t* m, const float* W, int32_t B, const float* pts, int64_t N, float scale,
                         double* out_sum) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t b = 0; b < B; ++b) {
        double acc = 0.0;
        for (int64_t n = 0; n < N; ++n) {
            float x[3];
            affine_apply(W + 16 * (int64_t)b, pts + 3 * n, x); /* :81-82 */
            const float sd = scale * mesh_distance(m, x);        /* :92 */
            acc += (double)(sd * sd);
        }
        out_sum[b] = acc;
    }
}

/* batch_chamfer_dist, chamfer.py:84-85, obj_sdf = CachedSDF branch */
void oracle_chamfer_grid(const oracle_grid_t* g, const float* W, int32_t B, const float* pts, int64_t N, float scale,
                         double* out_sum) {
#pragma omp parallel for schedule(static)
    for (int32_t b = 0; b < B; ++b) {
        double acc = 0.0;
        for (int64_t n = 0; n < N; ++n) {
            float x[3], v, gr[3];
            affine_apply(W + 16 * (int64_t)b, pts + 3 * n, x);
            cached_lookup(g, x, &v, gr);
            const float sd = scale * v;
            acc += (double)(sd * sd);
        }
        out_sum[b] = acc;
    }
}

/* C = A @ B for row-major 4x4, k-ordered fma chain (the f32 MFMA's rounding sequence) */
static void matmul4(const float* A, const float* B, float* C) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = A[4 * i] * B[j];
            for (int k = 1; k < 4; ++k) acc = fmaf(A[4 * i + k], B[4 * k + j], acc);
            C[4 * i + j] = acc;
        }
}

/* rigid inverse: [R t; 0 1]^-1 = [R^T  -R^T t; 0 1] */
static void rigid_inverse(const float* M, float* Mi) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Mi[4 * i + j] = M[4 * j + i];
        Mi[4 * i + 3] = -fmaf(M[8 + i], M[11], fmaf(M[4 + i], M[7], M[i] * M[3]));
    }
    Mi[12] = Mi[13] = Mi[14] = 0.f;
    Mi[15] = 1.f;
}

/* model_to_sdf.py:104-113: obj_to_link[s*A+a] = offset_inv[s] @ inverse(link_world[s*A+a]) */
void oracle_transform_stack(const float* offset_inv, const float* link_world, int32_t S, int32_t A, float* out) {
    for (int32_t s = 0; s < S; ++s)
        for (int32_t a = 0; a < A; ++a) {
            float inv[16];
            rigid_inverse(link_world + 16 * ((int64_t)s * A + a), inv);
            matmul4(offset_inv + 16 * (int64_t)s, inv, out + 16 * ((int64_t)s * A + a));
        }
}

/* ------------------------------------------------------------------------------------------------------------
 * Forward kinematics (model_to_sdf.py:94-102; pytorch_kinematics Chain.forward_kinematics, absent here: restated as
 * world[f] = world[parent] @ origin[f] @ motion(joint, q) with Rodrigues' rotation for revolute joints).
 * Same operation sequence as csrc/fk.hip.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct oracle_joint {
    int32_t parent, jtype, jcol, leaf_slot;
    float axis[3], reserved;
    float origin[12];
} oracle_joint_t;

static void compose_affine(const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = A[4 * i] * B[j];
            acc = fmaf(A[4 * i + 1], B[4 + j], acc);
            acc = fmaf(A[4 * i + 2], B[8 + j], acc);
            if (j == 3) acc = acc + A[4 * i + 3];
            C[4 * i + j] = acc;
        }
}

/* world: [F][A][12] (rows 0..2), link_world: [S*A][16] leaf-major */
void oracle_chain_fk(const oracle_joint_t* joints, int32_t F, const float* q, const float* sin_q, const float* cos_q,
                     int32_t A, int32_t M, float* world, float* link_world) {
    for (int32_t a = 0; a < A; ++a)
        for (int32_t f = 0; f < F; ++f) {
            const oracle_joint_t* J = &joints[f];
            float P[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, m[12], out[12];
            if (J->parent >= 0) memcpy(P, world + 12 * ((int64_t)J->parent * A + a), sizeof(P));
            compose_affine(P, J->origin, m);
            if (J->jtype == 1) {
                const float s = sin_q[(int64_t)a * M + J->jcol], c = cos_q[(int64_t)a * M + J->jcol];
                const float x = J->axis[0], y = J->axis[1], z = J->axis[2];
                const float t = 1.f - c, tx = t * x, ty = t * y, tz = t * z;
                const float R[12] = {fmaf(tx, x, c),        fmaf(tx, y, -(s * z)), fmaf(tx, z, s * y),    0.f,
                                     fmaf(tx, y, s * z),    fmaf(ty, y, c),        fmaf(ty, z, -(s * x)), 0.f,
                                     fmaf(tx, z, -(s * y)), fmaf(ty, z, s * x),    fmaf(tz, z, c),        0.f};
                compose_affine(m, R, out);
                memcpy(m, out, sizeof(m));
            } else if (J->jtype == 2) {
                const float d = q[(int64_t)a * M + J->jcol];
                const float T[12] = {1, 0, 0, J->axis[0] * d, 0, 1, 0, J->axis[1] * d, 0, 0, 1, J->axis[2] * d};
                compose_affine(m, T, out);
                memcpy(m, out, sizeof(m));
            }
            memcpy(world + 12 * ((int64_t)f * A + a), m, sizeof(m));
            if (J->leaf_slot >= 0) {
                float* o = link_world + 16 * ((int64_t)J->leaf_slot * A + a);
                memcpy(o, m, sizeof(m));
                o[12] = o[13] = o[14] = 0.f;
                o[15] = 1.f;
            }
        }
}
