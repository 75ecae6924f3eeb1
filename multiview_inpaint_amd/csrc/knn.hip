// distCUDA2 of the simple-knn plug-in the reference imports (gs-simp/scene/gaussian_model.py:20; call sites :134, :546,
// :623): for every point the MEAN of the squared distances to its 3 nearest other points. Third-party CUDA package,
// absent from the reference tree (SURVEY.md §8b "second boundary symbol"); published behaviour restated: exact 3-NN,
// the point itself excluded by index, (d0 + d1 + d2) / 3, FLT_MAX entries when fewer than 3 other points exist.
// Not on the per-step path (initialisation and densification only), so: exact tiled brute force. 256 queries per block
// in registers, candidates through LDS 256 at a time, branch-free sorted insertion (5 min/max per pair).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_train_ops.h"

namespace mvi {

int train_fail(int code, const char* msg);

__global__ __launch_bounds__(256) void knn3_mean_dist2_kernel(const float* __restrict__ pts, int N, float* __restrict__ out) {
    __shared__ float s_x[256], s_y[256], s_z[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < N;
    const float qx = live ? pts[3 * (size_t)i] : 0.f, qy = live ? pts[3 * (size_t)i + 1] : 0.f, qz = live ? pts[3 * (size_t)i + 2] : 0.f;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    for (int j0 = 0; j0 < N; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        s_x[threadIdx.x] = j < N ? pts[3 * (size_t)j] : 0.f;
        s_y[threadIdx.x] = j < N ? pts[3 * (size_t)j + 1] : 0.f;
        s_z[threadIdx.x] = j < N ? pts[3 * (size_t)j + 2] : 0.f;
        __syncthreads();
        const int n = min(256, N - j0);
#pragma unroll 8
        for (int k = 0; k < n; ++k) {
            const float dx = qx - s_x[k], dy = qy - s_y[k], dz = qz - s_z[k];
            float d = dx * dx + dy * dy + dz * dz;
            d = (j0 + k == i) ? FLT_MAX : d;                 // the point itself is skipped by index
            const float n2 = fminf(b2, fmaxf(b1, d));
            const float n1 = fminf(b1, fmaxf(b0, d));
            b0 = fminf(b0, d); b1 = n1; b2 = n2;
        }
    }
    if (live) out[i] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace mvi

extern "C" int mvi_knn3_mean_dist2(const float* points, int32_t N, float* mean_dist2, void* stream) {
    if (N < 0) return mvi::train_fail(MVI_EINVAL, "knn3_mean_dist2: negative point count");
    if (N == 0) return MVI_OK;
    if (!points || !mean_dist2) return mvi::train_fail(MVI_EINVAL, "knn3_mean_dist2: NULL pointer");
    hipLaunchKernelGGL(mvi::knn3_mean_dist2_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, points, N, mean_dist2);
    return hipGetLastError() == hipSuccess ? MVI_OK : mvi::train_fail(MVI_EHIP, "knn3_mean_dist2: kernel launch failed");
}
