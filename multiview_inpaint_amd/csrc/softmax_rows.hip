// In-place scaled row softmax for gfx950: x[r, :] = softmax(scale * x[r, :]).
// Used by the first-stage autoencoder's single-head attention (svd_inpaint1/sgm/modules/diffusionmodules/model.py:180-195:
// scaled_dot_product_attention with one head of D = C = 512), where the score matrix of a frame chunk is
// materialised by a library GEMM — with 288 GB of HBM an S x S fp32 matrix (340 MB per frame at 72x128 latents) is
// cheap to hold, and the D = 512 contraction is a plain GEMM. HBM-bound: one read + one write per score.
//
// One 256-thread block per row. The row goes through LDS as fp32 (16-byte global accesses), so global memory is
// touched once each way for rows up to kSmMaxLds elements; longer rows take the three-pass form (max, sum,
// normalise) straight from global memory. fp32 statistics; exp2 with the scale folded into one fma.
#include <hip/hip_runtime.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"
#include "unet_io.h"

namespace mvi {

int unet_fail(int code, const char* msg);

constexpr int kSmThreads = 256;
constexpr int kSmMaxLds = 12288;        // 48 KB of fp32 per block: 3 blocks per CU

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float t = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                    // s_red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) s_red[wave] = v;
    __syncthreads();
    float r = s_red[0];
#pragma unroll
    for (int w = 1; w < kSmThreads / 64; ++w) r = is_max ? fmaxf(r, s_red[w]) : r + s_red[w];
    return r;
}

template <typename T>
__global__ __launch_bounds__(kSmThreads) void softmax_rows_lds_kernel(T* __restrict__ x, int cols, float scale_log2e) {
    constexpr int V = Io<T>::kVec;
    extern __shared__ float s_row[];
    __shared__ float s_red[kSmThreads / 64];
    T* row = x + (int64_t)blockIdx.x * cols;
    const int nvec = cols / V;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nvec; i += kSmThreads) {
        float v[V];
        Io<T>::load(row + (int64_t)i * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) { s_row[i * V + j] = v[j]; m = fmaxf(m, v[j]); }
    }
    for (int i = nvec * V + threadIdx.x; i < cols; i += kSmThreads) { float v = Io<T>::ld1(row + i); s_row[i] = v; m = fmaxf(m, v); }
    m = block_reduce(m, true, s_red);
    // scale > 0: max(scale * x) = scale * max(x)
    const float mb = m * scale_log2e;
    float sum = 0.f;
    for (int i = threadIdx.x; i < cols; i += kSmThreads) {
        float e = exp2f(fmaf(s_row[i], scale_log2e, -mb));
        s_row[i] = e;
        sum += e;
    }
    sum = block_reduce(sum, false, s_red);
    const float inv = 1.0f / sum;
    for (int i = threadIdx.x; i < nvec; i += kSmThreads) {
        float v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = s_row[i * V + j] * inv;
        Io<T>::store(row + (int64_t)i * V, v);
    }
    for (int i = nvec * V + threadIdx.x; i < cols; i += kSmThreads) Io<T>::st1(row + i, s_row[i] * inv);
}

template <typename T>
__global__ __launch_bounds__(kSmThreads) void softmax_rows_global_kernel(T* __restrict__ x, int cols, float scale_log2e) {
    __shared__ float s_red[kSmThreads / 64];
    T* row = x + (int64_t)blockIdx.x * cols;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < cols; i += kSmThreads) m = fmaxf(m, Io<T>::ld1(row + i));
    m = block_reduce(m, true, s_red);
    const float mb = m * scale_log2e;
    float sum = 0.f;
    for (int i = threadIdx.x; i < cols; i += kSmThreads) sum += exp2f(fmaf(Io<T>::ld1(row + i), scale_log2e, -mb));
    sum = block_reduce(sum, false, s_red);
    const float inv = 1.0f / sum;
    for (int i = threadIdx.x; i < cols; i += kSmThreads)
        Io<T>::st1(row + i, exp2f(fmaf(Io<T>::ld1(row + i), scale_log2e, -mb)) * inv);
}

template <typename T>
static int softmax_rows_launch(void* x, int64_t rows, int cols, float scale, hipStream_t st) {
    if (rows > 0x7FFFFFFFll) return MVI_EINVAL;
    const float sl = scale * 1.4426950408889634f;
    constexpr int V = Io<T>::kVec;
    // the vector path needs 16-byte aligned rows: cols a multiple of the vector width
    if (cols <= kSmMaxLds && cols % V == 0)
        hipLaunchKernelGGL((softmax_rows_lds_kernel<T>), dim3((unsigned)rows), dim3(kSmThreads), (size_t)cols * sizeof(float), st,
                           (T*)x, cols, sl);
    else
        hipLaunchKernelGGL((softmax_rows_global_kernel<T>), dim3((unsigned)rows), dim3(kSmThreads), 0, st, (T*)x, cols, sl);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi

extern "C" int mvi_softmax_rows(void* x, int64_t rows, int32_t cols, float scale, int32_t dtype, void* stream) {
    using namespace mvi;
    if (rows < 0 || cols <= 0) return unet_fail(MVI_EINVAL, "softmax_rows: bad shape");
    if (!(scale > 0.f)) return unet_fail(MVI_EINVAL, "softmax_rows: scale must be positive");
    if (rows == 0) return MVI_OK;
    if (!x) return unet_fail(MVI_EINVAL, "softmax_rows: NULL pointer");
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = softmax_rows_launch<float>(x, rows, cols, scale, (hipStream_t)stream); break;
        case MVI_DT_BF16: rc = softmax_rows_launch<__hip_bfloat16>(x, rows, cols, scale, (hipStream_t)stream); break;
        case MVI_DT_F16: rc = softmax_rows_launch<__half>(x, rows, cols, scale, (hipStream_t)stream); break;
        default: return unet_fail(MVI_EINVAL, "softmax_rows: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "softmax_rows: too many rows");
    return rc ? unet_fail(MVI_EHIP, "softmax_rows: kernel launch failed") : MVI_OK;
}
