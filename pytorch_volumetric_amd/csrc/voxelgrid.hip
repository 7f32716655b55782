// Dense voxel containers addressed by real-valued points (reference voxel.py:42-103 over TorchMultidimView): read the
// values at points, write values at points -- the scatter twin of the query kernels' gather, over the same index
// arithmetic (grid_lookup.h).  SURVEY.md 8(f) rank 4.  HBM-bound streaming kernels: 12 B read + 4 (1) B written or read
// per point plus one random 4 (1) B access into the grid.
#include "common.h"
#include "grid_lookup.h"

namespace pvamd {

template <bool F64, typename T>
__global__ __launch_bounds__(256) void voxel_gather_kernel(pvamd_grid_t g, const T* __restrict__ storage,
                                                           const float* __restrict__ pts, int64_t P, T invalid,
                                                           T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        int flat;
        const bool valid = voxel_flat<F64>(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], flat);
        out[i] = valid ? storage[flat] : invalid;
    }
}

// who writes a voxel that several points fall into: the LAST point in input order, as a sequential loop (and numpy /
// torch-CPU index assignment) would leave it
template <bool F64>
__global__ __launch_bounds__(256) void voxel_owner_kernel(pvamd_grid_t g, const float* __restrict__ pts, int64_t P,
                                                          int* __restrict__ owner) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        int flat;
        if (voxel_flat<F64>(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], flat)) atomicMax(owner + flat, (int)i);
    }
}

template <bool F64, typename T>
__global__ __launch_bounds__(256) void voxel_scatter_kernel(pvamd_grid_t g, T* __restrict__ storage,
                                                            const float* __restrict__ pts, const T* __restrict__ values,
                                                            T scalar, int64_t P, const int* __restrict__ owner) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        int flat;
        if (!voxel_flat<F64>(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], flat)) continue;  // out of range: ignored
        if (values == nullptr) storage[flat] = scalar;  // every writer stores the same value: order is immaterial
        else if (owner[flat] == (int)i) storage[flat] = values[i];
    }
}

__global__ void fill_i32_kernel(int* p, int64_t n, int v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

template <typename T>
static int gather_impl(const pvamd_grid_t* grid, const T* storage, const float* points, int64_t P, T invalid, T* out,
                       void* stream) {
    if (P < 0) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grid || !storage || !points || !out) return PVAMD_E_NULL;
    if (int e = check_grid(*grid, /*need_vox=*/false)) return e;
    const dim3 gd(stream_grid(P, 256)), block(256);
    if (grid->index_f64) hipLaunchKernelGGL((voxel_gather_kernel<true, T>), gd, block, 0, (hipStream_t)stream, *grid, storage, points, P, invalid, out);
    else hipLaunchKernelGGL((voxel_gather_kernel<false, T>), gd, block, 0, (hipStream_t)stream, *grid, storage, points, P, invalid, out);
    return (int)hipGetLastError();
}

template <typename T>
static int scatter_impl(const pvamd_grid_t* grid, T* storage, const float* points, const T* values, T scalar, int64_t P,
                        int32_t* owner_scratch, void* stream) {
    if (P < 0 || P > (int64_t)INT32_MAX) return PVAMD_E_SHAPE;
    if (P == 0) return 0;
    if (!grid || !storage || !points) return PVAMD_E_NULL;
    if (values && !owner_scratch) return PVAMD_E_NULL;
    if (int e = check_grid(*grid, /*need_vox=*/false)) return e;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gd(stream_grid(P, 256)), block(256);
    const bool f64 = grid->index_f64 != 0;
    if (values) {
        const int64_t nvox = (int64_t)grid->shape[0] * grid->shape[1] * grid->shape[2];
        hipLaunchKernelGGL(fill_i32_kernel, dim3(stream_grid(nvox, 256)), block, 0, s, owner_scratch, nvox, -1);
        if (f64) hipLaunchKernelGGL((voxel_owner_kernel<true>), gd, block, 0, s, *grid, points, P, owner_scratch);
        else hipLaunchKernelGGL((voxel_owner_kernel<false>), gd, block, 0, s, *grid, points, P, owner_scratch);
    }
    if (f64) hipLaunchKernelGGL((voxel_scatter_kernel<true, T>), gd, block, 0, s, *grid, storage, points, values, scalar, P, owner_scratch);
    else hipLaunchKernelGGL((voxel_scatter_kernel<false, T>), gd, block, 0, s, *grid, storage, points, values, scalar, P, owner_scratch);
    return (int)hipGetLastError();
}

}  // namespace pvamd

using namespace pvamd;

extern "C" int pvamd_voxel_gather_f32(const pvamd_grid_t* grid, const float* storage, const float* points, int64_t P,
                                      float invalid_value, float* out, void* stream) {
    return gather_impl<float>(grid, storage, points, P, invalid_value, out, stream);
}
extern "C" int pvamd_voxel_gather_u8(const pvamd_grid_t* grid, const uint8_t* storage, const float* points, int64_t P,
                                     uint8_t invalid_value, uint8_t* out, void* stream) {
    return gather_impl<uint8_t>(grid, storage, points, P, invalid_value, out, stream);
}
extern "C" int pvamd_voxel_scatter_f32(const pvamd_grid_t* grid, float* storage, const float* points, const float* values,
                                       float scalar, int64_t P, int32_t* owner_scratch, void* stream) {
    return scatter_impl<float>(grid, storage, points, values, scalar, P, owner_scratch, stream);
}
extern "C" int pvamd_voxel_scatter_u8(const pvamd_grid_t* grid, uint8_t* storage, const float* points,
                                      const uint8_t* values, uint8_t scalar, int64_t P, int32_t* owner_scratch,
                                      void* stream) {
    return scatter_impl<uint8_t>(grid, storage, points, values, scalar, P, owner_scratch, stream);
}
