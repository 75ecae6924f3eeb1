// out[n, c, p] = h[n, c, p] + bias[c] (+ x[n, c, p]) for gfx950 — one pass over NC(T)HW activations.
// PyTorch-ROCm adds a convolution's bias as a separate broadcast kernel after the MIOpen kernel, and the
// ResBlock then adds the skip tensor in yet another pass (svd_inpaint1/sgm/modules/diffusionmodules/
// openaimodel.py:354 `self.skip_connection(x) + h`). The modules call their convolutions without bias and
// fold it here (or into the following GroupNorm's chan_bias). HBM-bound, 16 B per lane.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {
int unet_fail(int code, const char* msg);

template <typename T> struct BVec;
template <> struct BVec<float> {
    static constexpr int N = 4;
    __device__ static void load(const float* p, float* o) { float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    __device__ static void store(float* p, const float* o) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
    __device__ static float ld1(const float* p) { return *p; }
    __device__ static void st1(float* p, float v) { *p = v; }
};
template <> struct BVec<__hip_bfloat16> {
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
    __device__ static float ld1(const __hip_bfloat16* p) { return __bfloat162float(*p); }
    __device__ static void st1(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};
template <> struct BVec<__half> {
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
    __device__ static float ld1(const __half* p) { return __half2float(*p); }
    __device__ static void st1(__half* p, float v) { *p = __float2half(v); }
};

// alpha (optional, [N] fp32): out = x + (1 - alpha[n]) * (h + bias) — the AlphaBlender of a VideoResBlock applied to the
// temporal ResBlock's tail, alpha * x + (1 - alpha) * (x + h + bias), without materialising x + h + bias.
template <typename T, bool VEC, bool SILU = false>
__global__ __launch_bounds__(256) void bias_residual_kernel(const T* __restrict__ h, const T* __restrict__ x,
                                                            const float* __restrict__ bias, T* __restrict__ out,
                                                            int64_t total, int C, int64_t S,
                                                            const float* __restrict__ alpha = nullptr) {
    constexpr int N = BVec<T>::N;
    if (VEC) {
        const int64_t nvec = total / N;
        for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
            const int64_t e = v * N;
            const int64_t nc = e / S;                                  // S % N == 0: one channel per vector
            const float b = bias ? bias[nc % C] : 0.0f;
            const float wgt = alpha ? 1.0f - alpha[nc / C] : 1.0f;
            float a[N];
            BVec<T>::load(h + e, a);
#pragma unroll
            for (int k = 0; k < N; ++k) a[k] = alpha ? wgt * (a[k] + b) : a[k];
            if (x) {
                float r[N];
                BVec<T>::load(x + e, r);
#pragma unroll
                for (int k = 0; k < N; ++k) a[k] += r[k];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (!alpha) a[k] += b;
                if (SILU) a[k] = a[k] * __builtin_amdgcn_rcpf(1.0f + __expf(-a[k]));
            }
            BVec<T>::store(out + e, a);
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int64_t nc = e / S;
            float a = BVec<T>::ld1(h + e) + (bias ? bias[nc % C] : 0.0f);
            if (alpha) a *= 1.0f - alpha[nc / C];
            if (x) a += BVec<T>::ld1(x + e);
            if (SILU) a = a * __builtin_amdgcn_rcpf(1.0f + __expf(-a));
            BVec<T>::st1(out + e, a);
        }
    }
}

template <typename T, bool SILU = false>
static int bias_residual_launch(const void* h, const void* x, const float* bias, void* out, int64_t total, int C, int64_t S,
                                hipStream_t st, const float* alpha = nullptr) {
    constexpr int N = BVec<T>::N;
    const bool vec = (S % N == 0) && (((uintptr_t)h | (uintptr_t)x | (uintptr_t)out) % 16 == 0);
    int64_t work = vec ? total / N : total;
    int64_t blocks = (work + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks < 1) blocks = 1;
    if (vec)
        hipLaunchKernelGGL((bias_residual_kernel<T, true, SILU>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)h, (const T*)x, bias, (T*)out, total, C, S, alpha);
    else
        hipLaunchKernelGGL((bias_residual_kernel<T, false, SILU>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)h, (const T*)x, bias, (T*)out, total, C, S, alpha);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
// out[n] = (h[n] | skip[n] + ctrl[n]) along channels: the decoder's `torch.cat([h, hs.pop() + control.pop()], dim=1)`
// (models/csvd.py:79-91) without materialising the sum. blockIdx.y = sample; lanes walk the output row of that sample.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void concat_add_kernel(const T* __restrict__ h, const T* __restrict__ skip,
                                                         const T* __restrict__ ctrl, T* __restrict__ out,
                                                         int64_t n1, int64_t n2) {      // n1 = C1 * S, n2 = C2 * S
    constexpr int N = VEC ? BVec<T>::N : 1;
    const int64_t n = blockIdx.y;
    const T* __restrict__ hp = h + n * n1;
    const T* __restrict__ sp = skip + n * n2;
    const T* __restrict__ cp = ctrl ? ctrl + n * n2 : nullptr;
    T* __restrict__ op = out + n * (n1 + n2);
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * N; e < n1 + n2; e += (int64_t)gridDim.x * 256 * N) {
        if (VEC) {
            float a[BVec<T>::N];
            if (e < n1) {
                BVec<T>::load(hp + e, a);
            } else {
                BVec<T>::load(sp + (e - n1), a);
                if (cp) {
                    float r[BVec<T>::N];
                    BVec<T>::load(cp + (e - n1), r);
#pragma unroll
                    for (int k = 0; k < BVec<T>::N; ++k) a[k] += r[k];
                }
            }
            BVec<T>::store(op + e, a);
        } else {
            float a = e < n1 ? BVec<T>::ld1(hp + e) : BVec<T>::ld1(sp + (e - n1)) + (cp ? BVec<T>::ld1(cp + (e - n1)) : 0.0f);
            BVec<T>::st1(op + e, a);
        }
    }
}

template <typename T>
static int concat_add_launch(const void* h, const void* skip, const void* ctrl, void* out, int64_t N_, int64_t n1, int64_t n2,
                             hipStream_t st) {
    constexpr int V = BVec<T>::N;
    const bool vec = n1 % V == 0 && n2 % V == 0 &&
                     (((uintptr_t)h | (uintptr_t)skip | (uintptr_t)ctrl | (uintptr_t)out) % 16 == 0);
    int64_t work = vec ? (n1 + n2) / V : n1 + n2;
    int64_t bx = (work + 255) / 256;
    if (bx > 4096) bx = 4096;
    if (bx < 1) bx = 1;
    const dim3 grid((unsigned)bx, (unsigned)N_);
    if (vec)
        hipLaunchKernelGGL((concat_add_kernel<T, true>), grid, dim3(256), 0, st, (const T*)h, (const T*)skip, (const T*)ctrl, (T*)out, n1, n2);
    else
        hipLaunchKernelGGL((concat_add_kernel<T, false>), grid, dim3(256), 0, st, (const T*)h, (const T*)skip, (const T*)ctrl, (T*)out, n1, n2);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
}  // namespace mvi

extern "C" int mvi_concat_add(const void* h, const void* skip, const void* ctrl, void* out, int64_t N, int32_t C1, int32_t C2,
                              int64_t spatial, int32_t dtype, void* stream) {
    if (N < 0 || N > 65535 || C1 < 0 || C2 < 0 || spatial < 0) return mvi::unet_fail(MVI_EINVAL, "concat_add: bad shape (N must be below 65536)");
    const int64_t n1 = (int64_t)C1 * spatial, n2 = (int64_t)C2 * spatial;
    if (N == 0 || n1 + n2 == 0) return MVI_OK;
    if ((n1 && !h) || (n2 && !skip) || !out) return mvi::unet_fail(MVI_EINVAL, "concat_add: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::concat_add_launch<float>(h, skip, ctrl, out, N, n1, n2, st); break;
        case MVI_DT_BF16: rc = mvi::concat_add_launch<__hip_bfloat16>(h, skip, ctrl, out, N, n1, n2, st); break;
        case MVI_DT_F16: rc = mvi::concat_add_launch<__half>(h, skip, ctrl, out, N, n1, n2, st); break;
        default: return mvi::unet_fail(MVI_EINVAL, "concat_add: unknown dtype");
    }
    return rc ? mvi::unet_fail(MVI_EHIP, "concat_add: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_bias_residual_add(const void* h, const void* x, const float* bias, void* out, int64_t N, int32_t C,
                                     int64_t spatial, int32_t dtype, void* stream) {
    if (N < 0 || C <= 0 || spatial < 0) return mvi::unet_fail(MVI_EINVAL, "bias_residual_add: bad shape");
    const int64_t total = N * C * spatial;
    if (total == 0) return MVI_OK;
    if (!h || !out) return mvi::unet_fail(MVI_EINVAL, "bias_residual_add: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::bias_residual_launch<float>(h, x, bias, out, total, C, spatial, st); break;
        case MVI_DT_BF16: rc = mvi::bias_residual_launch<__hip_bfloat16>(h, x, bias, out, total, C, spatial, st); break;
        case MVI_DT_F16: rc = mvi::bias_residual_launch<__half>(h, x, bias, out, total, C, spatial, st); break;
        default: return mvi::unet_fail(MVI_EINVAL, "bias_residual_add: unknown dtype");
    }
    return rc ? mvi::unet_fail(MVI_EHIP, "bias_residual_add: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_bias_residual_blend(const void* h, const void* x, const float* bias, const float* alpha, void* out, int64_t N,
                                       int32_t C, int64_t spatial, int32_t dtype, void* stream) {
    if (N < 0 || C <= 0 || spatial < 0) return mvi::unet_fail(MVI_EINVAL, "bias_residual_blend: bad shape");
    const int64_t total = N * C * spatial;
    if (total == 0) return MVI_OK;
    if (!h || !x || !alpha || !out) return mvi::unet_fail(MVI_EINVAL, "bias_residual_blend: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::bias_residual_launch<float>(h, x, bias, out, total, C, spatial, st, alpha); break;
        case MVI_DT_BF16: rc = mvi::bias_residual_launch<__hip_bfloat16>(h, x, bias, out, total, C, spatial, st, alpha); break;
        case MVI_DT_F16: rc = mvi::bias_residual_launch<__half>(h, x, bias, out, total, C, spatial, st, alpha); break;
        default: return mvi::unet_fail(MVI_EINVAL, "bias_residual_blend: unknown dtype");
    }
    return rc ? mvi::unet_fail(MVI_EHIP, "bias_residual_blend: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_bias_silu(const void* h, const float* bias, void* out, int64_t N, int32_t C, int64_t spatial, int32_t dtype,
                             void* stream) {
    if (N < 0 || C <= 0 || spatial < 0) return mvi::unet_fail(MVI_EINVAL, "bias_silu: bad shape");
    const int64_t total = N * C * spatial;
    if (total == 0) return MVI_OK;
    if (!h || !out) return mvi::unet_fail(MVI_EINVAL, "bias_silu: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::bias_residual_launch<float, true>(h, nullptr, bias, out, total, C, spatial, st); break;
        case MVI_DT_BF16: rc = mvi::bias_residual_launch<__hip_bfloat16, true>(h, nullptr, bias, out, total, C, spatial, st); break;
        case MVI_DT_F16: rc = mvi::bias_residual_launch<__half, true>(h, nullptr, bias, out, total, C, spatial, st); break;
        default: return mvi::unet_fail(MVI_EINVAL, "bias_silu: unknown dtype");
    }
    return rc ? mvi::unet_fail(MVI_EHIP, "bias_silu: kernel launch failed") : MVI_OK;
}
