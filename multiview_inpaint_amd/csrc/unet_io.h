// 16-byte vector I/O of the activation dtypes (fp32 / bf16 / f16) with fp32 arithmetic in between; shared by the
// elementwise and normalisation kernels of the denoise loop.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvi {

template <typename T> struct Io;
template <> struct Io<float> {
    static constexpr int kVec = 4;                        // elements per 16 B
    __device__ static void load(const float* p, float* o) { float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    __device__ static void store(float* p, const float* o) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
    __device__ static float ld1(const float* p) { return *p; }
    __device__ static void st1(float* p, float v) { *p = v; }
};
template <> struct Io<__hip_bfloat16> {
    static constexpr int kVec = 8;
    __device__ static void load(const __hip_bfloat16* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
    }
    __device__ static void store(__hip_bfloat16* p, const float* o) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __hip_bfloat16 a = __float2bfloat16(o[2 * i]), b = __float2bfloat16(o[2 * i + 1]);
            w[i] = (uint32_t) * reinterpret_cast<uint16_t*>(&a) | ((uint32_t) * reinterpret_cast<uint16_t*>(&b) << 16);
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static float ld1(const __hip_bfloat16* p) { return __bfloat162float(*p); }
    __device__ static void st1(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};
template <> struct Io<__half> {
    static constexpr int kVec = 8;
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

// value after a round trip through the storage type: what a separately materialised intermediate would hold
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }
template <> __device__ __forceinline__ float round_to<__half>(float v) { return __half2float(__float2half(v)); }

}  // namespace mvi
