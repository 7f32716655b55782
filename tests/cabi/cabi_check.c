/* Plain-C host program over include/pvamd.h: no Python, no torch.  Builds a small voxel cache, runs
 * pvamd_grid_finalize / pvamd_pack_grid / pvamd_cached_query / pvamd_voxel_index on hipMalloc'ed buffers, then the composed
 * query three ways (pvamd_composed_query, its in-workgroup regrouping forced, pvamd_group_points + pvamd_composed_query_grouped
 * over scratch the host sized with pvamd_group_scratch_bytes), then a mesh it generates through pvamd_mesh_prepare / pvamd_mesh_query /
 * pvamd_mesh_query_unordered, and compares every output with the CPU oracle linked next to it.  Exit code 0 = bit-exact.  (tests/test_cabi_gpu.py builds and
 * runs it; it doubles as the example a non-Python host language would follow.) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pvamd.h"

#include "pvamd_oracle.h" /* the oracle's grid layout and entry points (oracle/) */

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define CKP(x) do { int r_ = (x); if (r_ != 0) { printf("pvamd error %d at line %d\n", r_, __LINE__); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f; }

int main(void) {
    if (pvamd_abi_version() != PVAMD_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
    const int shape[3] = {21, 17, 25};
    const int64_t n = (int64_t)shape[0] * shape[1] * shape[2], P = 100003; /* not a multiple of 256: tail path too */
    unsigned seed = 12345u;
    float* val = (float*)malloc(n * 4); float* grad = (float*)malloc(n * 12); float* pts = (float*)malloc(P * 12);
    for (int64_t i = 0; i < n; ++i) { val[i] = frand(&seed) - 0.3f; for (int d = 0; d < 3; ++d) grad[3 * i + d] = frand(&seed) - 0.5f; }
    pvamd_grid_t g; memset(&g, 0, sizeof g);
    oracle_grid_t o; memset(&o, 0, sizeof o);
    const double lo[3] = {-0.31, -0.22, 0.05}, res = 0.02;
    for (int d = 0; d < 3; ++d) {
        g.shape[d] = o.shape[d] = shape[d];
        g.dmin[d] = o.dmin[d] = lo[d]; g.dmax[d] = o.dmax[d] = lo[d] + res * (shape[d] - 1);
        g.dres[d] = o.dres[d] = (g.dmax[d] - g.dmin[d]) / (shape[d] - 1);
        g.fmin[d] = o.fmin[d] = (float)g.dmin[d]; g.fmax[d] = o.fmax[d] = (float)g.dmax[d];
        g.fres[d] = o.fres[d] = (g.fmax[d] - g.fmin[d]) / (float)(shape[d] - 1);
        g.bb_min[d] = o.bb_min[d] = (float)(lo[d] + 0.08); g.bb_max[d] = o.bb_max[d] = (float)(g.dmax[d] - 0.08);
        o.dbb_min[d] = o.bb_min[d]; o.dbb_max[d] = o.bb_max[d];
    }
    for (int64_t i = 0; i < P; ++i) for (int d = 0; d < 3; ++d)
        pts[3 * i + d] = (float)(g.dmin[d] - 0.06 + frand(&seed) * (g.dmax[d] - g.dmin[d] + 0.12));
    float *dval, *dgrad, *dvox, *dpts, *dov, *dog; int64_t *dkey; uint8_t* doob;
    CK(hipMalloc((void**)&dval, n * 4)); CK(hipMalloc((void**)&dgrad, n * 12)); CK(hipMalloc((void**)&dvox, n * 16));
    CK(hipMalloc((void**)&dpts, P * 12)); CK(hipMalloc((void**)&dov, P * 4)); CK(hipMalloc((void**)&dog, P * 12));
    CK(hipMalloc((void**)&dkey, P * 24)); CK(hipMalloc((void**)&doob, P));
    CK(hipMemcpy(dval, val, n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgrad, grad, n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpts, pts, P * 12, hipMemcpyHostToDevice));
    float* ov = (float*)malloc(P * 4); float* og = (float*)malloc(P * 12); uint8_t* oo = (uint8_t*)malloc(P);
    float* hv = (float*)malloc(P * 4); float* hg = (float*)malloc(P * 12); uint8_t* ho = (uint8_t*)malloc(P);
    int64_t* okey = (int64_t*)malloc(P * 24); int64_t* hkey = (int64_t*)malloc(P * 24);
    int bad = 0;
    for (int f64 = 0; f64 <= 1; ++f64) for (int mode = 0; mode <= 1; ++mode) {
        g.vox = dvox; g.index_f64 = o.index_f64 = f64; g.oob_mode = o.oob_mode = mode; g.finalized = 0;
        o.val = val; o.grad = grad;
        CKP(pvamd_grid_finalize(&g));
        CKP(pvamd_pack_grid(dval, dgrad, n, dvox, NULL));
        CKP(pvamd_cached_query(&g, dpts, P, dov, dog, doob, NULL));
        CKP(pvamd_voxel_index(&g, dpts, P, dkey, NULL, NULL, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hv, dov, P * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hg, dog, P * 12, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ho, doob, P, hipMemcpyDeviceToHost)); CK(hipMemcpy(hkey, dkey, P * 24, hipMemcpyDeviceToHost));
        oracle_cached_query(&o, pts, P, ov, og, oo);
        oracle_voxel_index(&o, pts, P, okey, NULL, NULL);
        int64_t mism = 0, noob = 0;
        for (int64_t i = 0; i < P; ++i) {
            noob += oo[i];
            if (memcmp(&hv[i], &ov[i], 4) || memcmp(&hg[3 * i], &og[3 * i], 12) || ho[i] != oo[i] ||
                memcmp(&hkey[3 * i], &okey[3 * i], 24)) {
                /* NaN gradients (point inside the bounding box but outside the range) compare by bits too */
                if (!(isnan(hg[3 * i]) && isnan(og[3 * i]) && hv[i] == ov[i] && ho[i] == oo[i])) ++mism;
            }
        }
        printf("index_f64=%d oob_mode=%d: %lld points, %lld out of range, %lld mismatches\n", f64, mode, (long long)P,
               (long long)noob, (long long)mism);
        bad += mism != 0;
    }
    /* ---- ComposedSDF.__call__ (sdf.py:392-433): S leaves sharing the cache above under S x A rigid transforms ----
     * pvamd_composed_query as it picks its kernel, the same with the in-workgroup regrouping forced, and the pre-pass pair
     * pvamd_group_points + pvamd_composed_query_grouped over caller-provided scratch: all three against the oracle. */
    {
        enum { S = 5, A = 3 };
        const int64_t Pc = 3 * pvamd_group_chunk_points() + 777; /* a ragged last chunk */
        g.index_f64 = o.index_f64 = 0; g.oob_mode = o.oob_mode = PVAMD_OOB_BOUNDING_BOX; g.finalized = 0;
        CKP(pvamd_grid_finalize(&g));
        pvamd_grid_t gs[S]; oracle_grid_t os[S];
        for (int s2 = 0; s2 < S; ++s2) { gs[s2] = g; os[s2] = o; }
        float tf[S * A * 16];
        for (int k = 0; k < S * A; ++k) { /* rotation about a seeded axis (Rodrigues) + translation, row-major 4x4 */
            float ax[3] = {frand(&seed) - 0.5f, frand(&seed) - 0.5f, frand(&seed) - 0.5f};
            const float nrm = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-9f, th = 3.0f * frand(&seed);
            for (int d = 0; d < 3; ++d) ax[d] /= nrm;
            const float c = cosf(th), sn = sinf(th), K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
            float* M = tf + 16 * k; memset(M, 0, 64); M[15] = 1.f;
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q)
                M[4 * r + q] = (r == q ? c : 0.f) + sn * K[3 * r + q] + (1.f - c) * ax[r] * ax[q];
            for (int r = 0; r < 3; ++r) M[4 * r + 3] = 0.4f * (frand(&seed) - 0.5f);
        }
        float* cp = (float*)malloc(Pc * 12);
        for (int64_t i = 0; i < 3 * Pc; ++i) cp[i] = 1.2f * (frand(&seed) - 0.5f);
        float *dcp, *dtf, *dcv, *dcg; int32_t* dcl; pvamd_grid_t* dgs; void* dscr;
        const int64_t scr = pvamd_group_scratch_bytes(Pc);
        CK(hipMalloc((void**)&dcp, Pc * 12)); CK(hipMalloc((void**)&dtf, sizeof tf)); CK(hipMalloc((void**)&dgs, sizeof gs));
        CK(hipMalloc((void**)&dcv, A * Pc * 4)); CK(hipMalloc((void**)&dcg, A * Pc * 12)); CK(hipMalloc((void**)&dcl, A * Pc * 4));
        CK(hipMalloc(&dscr, scr));
        CK(hipMemcpy(dcp, cp, Pc * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(dtf, tf, sizeof tf, hipMemcpyHostToDevice));
        CK(hipMemcpy(dgs, gs, sizeof gs, hipMemcpyHostToDevice));
        float* rv = (float*)malloc(A * Pc * 4); float* rg = (float*)malloc(A * Pc * 12); int32_t* rl = (int32_t*)malloc(A * Pc * 4);
        float* cv = (float*)malloc(A * Pc * 4); float* cg = (float*)malloc(A * Pc * 12); int32_t* cl = (int32_t*)malloc(A * Pc * 4);
        oracle_composed_query(os, S, tf, A, cp, Pc, rv, rg, rl);
        const char* names[3] = {"pvamd_composed_query", "pvamd_composed_query, regrouping forced", "pvamd_group_points + pvamd_composed_query_grouped"};
        for (int way = 0; way < 3; ++way) {
            CK(hipMemset(dcv, 0xff, A * Pc * 4)); CK(hipMemset(dcg, 0xff, A * Pc * 12)); CK(hipMemset(dcl, 0xff, A * Pc * 4));
            if (way < 2) {
                CKP(pvamd_composed_query(dgs, S, dtf, A, dcp, Pc, dcv, dcg, dcl, way == 1 ? PVAMD_COMPOSED_FORCE_FUSED : 0, NULL));
            } else {
                CKP(pvamd_group_points(dcp, Pc, dscr, NULL));
                CKP(pvamd_composed_query_grouped(dgs, S, dtf, A, dscr, Pc, dcv, dcg, dcl, 0, NULL));
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(cv, dcv, A * Pc * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cg, dcg, A * Pc * 12, hipMemcpyDeviceToHost));
            CK(hipMemcpy(cl, dcl, A * Pc * 4, hipMemcpyDeviceToHost));
            int64_t mism = 0, inside = 0;
            for (int64_t i = 0; i < A * Pc; ++i) {
                inside += rv[i] < 0.f;
                if (memcmp(&cv[i], &rv[i], 4) || cl[i] != rl[i]) {
                    if (mism < 3) printf("  [%lld] val %.9g vs %.9g, leaf %d vs %d, grad (%g %g %g) vs (%g %g %g)\n", (long long)i, cv[i], rv[i],
                                         cl[i], rl[i], cg[3 * i], cg[3 * i + 1], cg[3 * i + 2], rg[3 * i], rg[3 * i + 1], rg[3 * i + 2]);
                    ++mism; continue;
                }
                for (int d = 0; d < 3; ++d) /* a NaN gradient (0 / 0 inside the box) compares as NaN == NaN */
                    if (memcmp(&cg[3 * i + d], &rg[3 * i + d], 4) && !(isnan(cg[3 * i + d]) && isnan(rg[3 * i + d]))) { ++mism; break; }
            }
            printf("%s: %d leaves x %d configurations x %lld points, %lld negative, %lld mismatches\n", names[way], S, A, (long long)Pc,
                   (long long)inside, (long long)mism);
            bad += mism != 0;
        }
    }
    /* ---- ObjectFactory._do_object_frame_closest_point (sdf.py:122-172): a 528-triangle ellipsoid built right here, prepared by
     * pvamd_mesh_prepare, queried by pvamd_mesh_query (caller order, with scratch) and pvamd_mesh_query_unordered ---- */
    {
        enum { NLON = 24, NLAT = 12, F = 2 * NLON * (NLAT - 1) };
        const double kPi = 3.14159265358979323846;
        float* tri = (float*)malloc(F * 36); float* nrm = (float*)malloc(F * 12);
        int f = 0;
        #define VERT(i, j, out) do { const double th = kPi * (i) / NLAT, ph = 2.0 * kPi * ((j) % NLON) / NLON; \
            (out)[0] = (float)(0.30 * sin(th) * cos(ph)); (out)[1] = (float)(0.20 * sin(th) * sin(ph)); (out)[2] = (float)(0.12 * cos(th) + 0.05); } while (0)
        for (int i = 0; i < NLAT; ++i) for (int j = 0; j < NLON; ++j) {
            float a[3], b[3], c[3], d[3];
            VERT(i, j, a); VERT(i + 1, j, b); VERT(i + 1, j + 1, c); VERT(i, j + 1, d);
            if (i > 0) { memcpy(tri + 9 * f, a, 12); memcpy(tri + 9 * f + 3, b, 12); memcpy(tri + 9 * f + 6, d, 12); ++f; }
            if (i < NLAT - 1) { memcpy(tri + 9 * f, b, 12); memcpy(tri + 9 * f + 3, c, 12); memcpy(tri + 9 * f + 6, d, 12); ++f; }
        }
        if (f != F) { printf("mesh construction: %d triangles, expected %d\n", f, (int)F); return 5; }
        for (int k = 0; k < F; ++k) { /* unit face normals in float64, rounded once (sdf.py:119-120) */
            const float* t = tri + 9 * k; double u[3], v[3], n[3];
            for (int d = 0; d < 3; ++d) { u[d] = (double)t[3 + d] - t[d]; v[d] = (double)t[6 + d] - t[d]; }
            n[0] = u[1] * v[2] - u[2] * v[1]; n[1] = u[2] * v[0] - u[0] * v[2]; n[2] = u[0] * v[1] - u[1] * v[0];
            const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int d = 0; d < 3; ++d) nrm[3 * k + d] = (float)(len > 0 ? n[d] / len : 0.0);
        }
        const int64_t Pm = 12000;
        float* mp = (float*)malloc(Pm * 12);
        for (int64_t i = 0; i < Pm; ++i) { mp[3 * i] = 0.9f * (frand(&seed) - 0.5f); mp[3 * i + 1] = 0.7f * (frand(&seed) - 0.5f); mp[3 * i + 2] = 0.5f * (frand(&seed) - 0.4f); }
        float *dtri, *dnrm, *drec, *dtiles, *dmp, *dq, *dd, *dgr, *dn; int32_t *drof, *dface, *dord; void* dms;
        CK(hipMalloc((void**)&dtri, F * 36)); CK(hipMalloc((void**)&dnrm, F * 12)); CK(hipMalloc((void**)&drec, PVAMD_REC_FLOATS(F) * 4));
        CK(hipMalloc((void**)&dtiles, PVAMD_TILES_FLOATS(F) * 4)); CK(hipMalloc((void**)&drof, F * 4)); CK(hipMalloc((void**)&dmp, Pm * 12));
        CK(hipMalloc((void**)&dq, Pm * 12)); CK(hipMalloc((void**)&dd, Pm * 4)); CK(hipMalloc((void**)&dgr, Pm * 12)); CK(hipMalloc((void**)&dn, Pm * 12));
        CK(hipMalloc((void**)&dface, Pm * 4)); CK(hipMalloc((void**)&dord, Pm * 4)); CK(hipMalloc(&dms, PVAMD_MESH_SCRATCH_BYTES(Pm)));
        CK(hipMemcpy(dtri, tri, F * 36, hipMemcpyHostToDevice)); CK(hipMemcpy(dnrm, nrm, F * 12, hipMemcpyHostToDevice));
        CK(hipMemcpy(dmp, mp, Pm * 12, hipMemcpyHostToDevice));
        CKP(pvamd_mesh_prepare(dtri, NULL, F, 1e-6f * (0.35f + 0.8f), drec, dtiles, drof, NULL));
        pvamd_mesh_t mesh; memset(&mesh, 0, sizeof mesh);
        mesh.normal = dnrm; mesh.rec = drec; mesh.tiles = dtiles; mesh.rec_of_face = drof; mesh.F = F;
        oracle_mesh_t om; memset(&om, 0, sizeof om);
        om.tri = tri; om.normal = nrm; om.F = F;
        const double far_corner[3] = {1.30, 1.20, 1.17}; /* bounding_box(padding=1.0)[:, 1] (sdf.py:147) */
        for (int d = 0; d < 3; ++d) mesh.ray_dir[d] = om.ray_dir[d] = far_corner[d];
        float* rq = (float*)malloc(Pm * 12); float* rd = (float*)malloc(Pm * 4); float* rgr = (float*)malloc(Pm * 12);
        float* rn = (float*)malloc(Pm * 12); int32_t* rf = (int32_t*)malloc(Pm * 4);
        float* hq = (float*)malloc(Pm * 12); float* hd = (float*)malloc(Pm * 4); float* hgr = (float*)malloc(Pm * 12);
        float* hn = (float*)malloc(Pm * 12); int32_t* hf = (int32_t*)malloc(Pm * 4);
        const uint64_t jitter = 20240607u;
        oracle_mesh_query(&om, mp, Pm, jitter, 0, rq, rd, rgr, rf, rn);
        for (int way = 0; way < 2; ++way) {
            CK(hipMemset(dd, 0xff, Pm * 4)); CK(hipMemset(dface, 0xff, Pm * 4));
            if (way == 0) CKP(pvamd_mesh_query(&mesh, dmp, NULL, Pm, jitter, 0, dq, dd, dgr, dface, dn, dms, NULL));
            else CKP(pvamd_mesh_query_unordered(&mesh, dmp, Pm, jitter, 0, dq, dd, dgr, dface, dn, dord, dms, NULL));
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hq, dq, Pm * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(hd, dd, Pm * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hgr, dgr, Pm * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(hn, dn, Pm * 12, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hf, dface, Pm * 4, hipMemcpyDeviceToHost));
            int64_t mism = 0, inside = 0;
            for (int64_t i = 0; i < Pm; ++i) {
                inside += rd[i] < 0.f;
                if (memcmp(&hd[i], &rd[i], 4) || hf[i] != rf[i] || memcmp(&hq[3 * i], &rq[3 * i], 12) || memcmp(&hgr[3 * i], &rgr[3 * i], 12) ||
                    memcmp(&hn[3 * i], &rn[3 * i], 12)) {
                    if (mism < 3) printf("  [%lld] d %.9g vs %.9g, face %d vs %d\n", (long long)i, hd[i], rd[i], hf[i], rf[i]);
                    ++mism;
                }
            }
            printf("%s: %d triangles, %lld points, %lld inside, %lld mismatches\n", way == 0 ? "pvamd_mesh_query" : "pvamd_mesh_query_unordered",
                   (int)F, (long long)Pm, (long long)inside, (long long)mism);
            bad += mism != 0 || inside == 0;
        }
        /* ---- batch_chamfer_dist (chamfer.py:79-94): B world->object transforms, the same points, mesh branch and cached-grid
         * branch; sums are float64 in an order of their own, so: relative 1e-9 (what the Python tests allow too) ---- */
        enum { B = 4 };
        float W[B * 16];
        memset(W, 0, sizeof W);
        for (int b = 0; b < B; ++b) { /* a small rotation about z and a shift */
            const float th = 0.1f * b, c = cosf(th), sn = sinf(th);
            float* M = W + 16 * b;
            M[0] = c; M[1] = -sn; M[4] = sn; M[5] = c; M[10] = 1.f; M[15] = 1.f;
            M[3] = 0.01f * b; M[7] = -0.02f * b; M[11] = 0.005f * b;
        }
        float* dW; double* dsum;
        CK(hipMalloc((void**)&dW, sizeof W)); CK(hipMalloc((void**)&dsum, B * 8));
        CK(hipMemcpy(dW, W, sizeof W, hipMemcpyHostToDevice));
        double want[B], got[B];
        oracle_chamfer_mesh(&om, W, B, mp, Pm, 1000.f, want);
        CKP(pvamd_chamfer_mesh(&mesh, dW, B, dmp, NULL, Pm, 1000.f, dsum, dms, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got, dsum, B * 8, hipMemcpyDeviceToHost));
        int64_t cm = 0;
        for (int b = 0; b < B; ++b) cm += !(fabs(got[b] - want[b]) <= 1e-9 * fabs(want[b]));
        printf("pvamd_chamfer_mesh: %d transforms x %lld points, sum[1] = %.12g vs %.12g, %lld mismatches\n", (int)B, (long long)Pm, got[1], want[1], (long long)cm);
        bad += cm != 0;
        g.index_f64 = o.index_f64 = 0; g.oob_mode = o.oob_mode = PVAMD_OOB_BOUNDING_BOX; g.finalized = 0;
        CKP(pvamd_grid_finalize(&g));
        oracle_chamfer_grid(&o, W, B, mp, Pm, 1000.f, want);
        CKP(pvamd_chamfer_grid(&g, dW, B, dmp, Pm, 1000.f, dsum, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got, dsum, B * 8, hipMemcpyDeviceToHost));
        cm = 0;
        for (int b = 0; b < B; ++b) cm += !(fabs(got[b] - want[b]) <= 1e-9 * fabs(want[b]));
        printf("pvamd_chamfer_grid: %d transforms x %lld points, sum[1] = %.12g vs %.12g, %lld mismatches\n", (int)B, (long long)Pm, got[1], want[1], (long long)cm);
        bad += cm != 0;
        /* ---- RobotSDF.set_joint_configuration's contraction (model_to_sdf.py:104-113) on the f32 MFMA: the B transforms above as
         * offsets, 4 x 3 of them as link poses ---- */
        enum { SS = 4, AA = 3 };
        float link[SS * AA * 16], stack_want[SS * AA * 16], stack_got[SS * AA * 16];
        for (int k = 0; k < SS * AA; ++k) {
            const float th = 0.37f * (k + 1), c = cosf(th), sn = sinf(th);
            float* M = link + 16 * k; memset(M, 0, 64);
            M[0] = c; M[2] = sn; M[5] = 1.f; M[8] = -sn; M[10] = c; M[15] = 1.f; M[3] = 0.1f * k; M[7] = 0.3f; M[11] = -0.05f * k;
        }
        float *dlink, *dstack;
        CK(hipMalloc((void**)&dlink, sizeof link)); CK(hipMalloc((void**)&dstack, sizeof link));
        CK(hipMemcpy(dlink, link, sizeof link, hipMemcpyHostToDevice));
        oracle_transform_stack(W, link, SS, AA, stack_want);
        CKP(pvamd_transform_stack(dW, dlink, SS, AA, dstack, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(stack_got, dstack, sizeof link, hipMemcpyDeviceToHost));
        const int64_t sm = memcmp(stack_got, stack_want, sizeof link) != 0;
        printf("pvamd_transform_stack: %d x %d matrices, %lld mismatches\n", (int)SS, (int)AA, (long long)sm);
        bad += sm != 0;
    }
    printf(bad ? "FAILED\n" : "C-ABI check passed: %s\n", pvamd_build_info());
    return bad ? 4 : 0;
}
