// Fused GEGLU for gfx950: out[r, j] = h[r, j] * gelu(h[r, inner + j]) (exact erf GELU), one pass.
// Replaces `x, gate = proj(x).chunk(2, -1); x * F.gelu(gate)`
// (svd_inpaint1/sgm/modules/attention.py:87-95), which PyTorch runs as a GELU over a strided view
// plus a strided multiply (3 full passes). HBM-bound: reads 2*inner, writes inner per row, 16 B/lane.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {
int unet_fail(int code, const char* msg);

__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752f)); }

template <typename T> struct GVec;
template <> struct GVec<float> {
    static constexpr int N = 4;
    __device__ static void load(const float* p, float* o) { float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    __device__ static void store(float* p, const float* o) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct GVec<__hip_bfloat16> {
    static constexpr int N = 8;
    __device__ static void load(const __hip_bfloat16* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
    }
    __device__ static void store(__hip_bfloat16* p, const float* o) {
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { f2 f = {o[2 * i], o[2 * i + 1]}; b2 r = __builtin_convertvector(f, b2); w[i] = *reinterpret_cast<uint32_t*>(&r); }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct GVec<__half> {
    static constexpr int N = 8;
    __device__ static void load(const __half* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        const __half2* h = reinterpret_cast<const __half2*>(&v);
#pragma unroll
        for (int i = 0; i < 4; ++i) { float2 f = __half22float2(h[i]); o[2 * i] = f.x; o[2 * i + 1] = f.y; }
    }
    __device__ static void store(__half* p, const float* o) {
        uint4 v;
        __half2* h = reinterpret_cast<__half2*>(&v);
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(o[2 * i], o[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = v;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* __restrict__ h, T* __restrict__ out, int64_t rows, int inner) {
    constexpr int N = GVec<T>::N;
    const int vec_per_row = inner / N;
    const int64_t total = rows * vec_per_row;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int64_t r = v / vec_per_row;
        const int j = (int)(v % vec_per_row) * N;
        float a[N], g[N];
        GVec<T>::load(h + r * 2 * inner + j, a);
        GVec<T>::load(h + r * 2 * inner + inner + j, g);
#pragma unroll
        for (int k = 0; k < N; ++k) a[k] *= gelu_erf(g[k]);
        GVec<T>::store(out + r * inner + j, a);
    }
}

template <typename T>
static int geglu_launch(const void* h, void* out, int64_t rows, int inner, hipStream_t st) {
    const int64_t total = rows * (inner / GVec<T>::N);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;             // grid-stride the rest
    hipLaunchKernelGGL((geglu_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)h, (T*)out, rows, inner);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
}  // namespace mvi

extern "C" int mvi_geglu(const void* h, void* out, int64_t rows, int32_t inner, int32_t dtype, void* stream) {
    if (rows < 0 || inner <= 0) return mvi::unet_fail(MVI_EINVAL, "geglu: bad shape");
    if (rows == 0) return MVI_OK;
    if (!h || !out) return mvi::unet_fail(MVI_EINVAL, "geglu: NULL pointer");
    const int n = dtype == MVI_DT_F32 ? 4 : 8;
    if (inner % n != 0 || ((uintptr_t)h | (uintptr_t)out) % 16 != 0)
        return mvi::unet_fail(MVI_EINVAL, "geglu: inner must be a multiple of the 16-byte vector and pointers 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::geglu_launch<float>(h, out, rows, inner, st); break;
        case MVI_DT_BF16: rc = mvi::geglu_launch<__hip_bfloat16>(h, out, rows, inner, st); break;
        case MVI_DT_F16: rc = mvi::geglu_launch<__half>(h, out, rows, inner, st); break;
        default: return mvi::unet_fail(MVI_EINVAL, "geglu: unknown dtype");
    }
    return rc ? mvi::unet_fail(MVI_EHIP, "geglu: kernel launch failed") : MVI_OK;
}
