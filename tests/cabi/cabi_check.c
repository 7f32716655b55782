/* Plain-C host program over include/pvamd.h: no Python, no torch.  Builds a small voxel cache, runs
 * pvamd_grid_finalize / pvamd_pack_grid / pvamd_cached_query / pvamd_voxel_index on hipMalloc'ed buffers and compares
 * every output with the CPU oracle linked next to it.  Exit code 0 = bit-exact.  (tests/test_cabi_gpu.py builds and
 * runs it; it doubles as the example a non-Python host language would follow.) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pvamd.h"

/* the oracle's grid layout (oracle/pvamd_oracle.c) */
typedef struct oracle_grid {
    const float* val; const float* grad;
    double dmin[3], dmax[3], dres[3];
    float fmin[3], fmax[3], fres[3];
    float bb_min[3], bb_max[3];
    int32_t shape[3]; int32_t index_f64; int32_t oob_mode; int32_t reserved;
} oracle_grid_t;
void oracle_cached_query(const oracle_grid_t*, const float*, int64_t, float*, float*, uint8_t*);
void oracle_voxel_index(const oracle_grid_t*, const float*, int64_t, int64_t*, int64_t*, uint8_t*);

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
    printf(bad ? "FAILED\n" : "C-ABI check passed: %s\n", pvamd_build_info());
    return bad ? 4 : 0;
}
