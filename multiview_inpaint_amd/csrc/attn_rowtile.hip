// fp32-math attention for gfx950: the validation-mode / short-sequence kernel.
// Replaces F.scaled_dot_product_attention / xformers.memory_efficient_attention
// (svd_inpaint1/sgm/modules/attention.py:332-336, :427-439) for
//   * fp32 inputs  — the 1e-4-parity validation mode (no reduced-precision operand anywhere),
//   * the temporal attention of the video transformer, S_q = S_k = T (14 or 25) with batch
//     (b h w) x heads in the tens of thousands: HBM-bound, one wave per problem,
//   * head dims the MFMA kernel is not built for.
// Layout: q/out [B, Sq, H, D], k/v [B, Sk, H, D] token-major (what the Linear layers produce), so no
// head-transpose copies exist on either side.
//
// One wave64 = 16 query rows x 4 lanes per row; each lane owns D/4 interleaved channels
// (d = 4*p + 16*i + c: a lane's 4-vectors are 16 B apart from its neighbours', so LDS reads of a
// K/V row are conflict-free b128). K/V tiles of up to 32 keys are staged in LDS per wave.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {

int unet_fail(int code, const char* msg);

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
// four consecutive elements (8-byte aligned for the 2-byte types, 16-byte for fp32) as floats
__device__ __forceinline__ void ld4(const float* p, float* o) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void ld4(const __hip_bfloat16* p, float* o) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
__device__ __forceinline__ void ld4(const __half* p, float* o) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    const __half2* h = reinterpret_cast<const __half2*>(&v);
    const float2 a = __half22float2(h[0]), b = __half22float2(h[1]);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
__device__ __forceinline__ void st4(float* p, const float* o) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
__device__ __forceinline__ void st4(__hip_bfloat16* p, const float* o) {
    const __hip_bfloat16 a = __float2bfloat16(o[0]), b = __float2bfloat16(o[1]), c = __float2bfloat16(o[2]), d = __float2bfloat16(o[3]);
    uint2 v;
    v.x = (uint32_t)*reinterpret_cast<const uint16_t*>(&a) | ((uint32_t)*reinterpret_cast<const uint16_t*>(&b) << 16);
    v.y = (uint32_t)*reinterpret_cast<const uint16_t*>(&c) | ((uint32_t)*reinterpret_cast<const uint16_t*>(&d) << 16);
    *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void st4(__half* p, const float* o) {
    uint2 v;
    __half2* h = reinterpret_cast<__half2*>(&v);
    h[0] = __floats2half2_rn(o[0], o[1]);
    h[1] = __floats2half2_rn(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = v;
}
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }

constexpr int kRtKeysMax = 32;   // keys per LDS tile (16 for short sequences: the temporal attention has Sk = T = 14)
constexpr int kRtWaves = 4;      // waves (= independent problems) per block

// DV = D / 16 : number of 4-channel vectors a lane owns (D in {16, 32, 64, 128})
// Addressing: batch index b = bo * inner + bi. Row r of (b, head h) starts at element
//   bo * outer_stride + bi * inner_stride + r * row_stride + h * D        (separately for q/out and k/v).
// Standard token-major [B, S, H, D]: inner = 1, outer_stride = S*H*D, row_stride = H*D.
// Temporal attention over x [(bo T), S, H, D] without regrouping tokens: one problem per (bo, s):
//   inner = S, inner_stride = H*D, outer_stride = T*S*H*D, row_stride = S*H*D, rows = frames.
struct RtLayout {
    int inner;
    int64_t q_outer, q_inner, q_row, k_outer, k_inner, k_row, o_outer, o_inner, o_row;
};

template <typename T, int DV, int kRtKeys>
__global__ __launch_bounds__(64 * kRtWaves) void attn_rowtile_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                     const T* __restrict__ v, T* __restrict__ out,
                                                                     int B, int H, int Sq, int Sk, float scale,
                                                                     int64_t n_problems, int q_tiles, RtLayout L) {
    constexpr int D = DV * 16;
    __shared__ __attribute__((aligned(16))) float s_k[kRtWaves][kRtKeys][D];
    __shared__ __attribute__((aligned(16))) float s_v[kRtWaves][kRtKeys][D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t prob = (int64_t)blockIdx.x * kRtWaves + wave;        // (b, h, q_tile)
    if (prob >= n_problems) return;                                     // whole wave exits together
    const int qt = (int)(prob % q_tiles);
    const int64_t bh = prob / q_tiles;
    const int h = (int)(bh % H);
    const int64_t b = bh / H;
    const int row = qt * 16 + (lane >> 2), p = lane & 3;
    const bool row_ok = row < Sq;
    const int64_t bo = b / L.inner, bi = b % L.inner;
    const int64_t qbase = bo * L.q_outer + bi * L.q_inner + (int64_t)h * D;
    const int64_t kbase = bo * L.k_outer + bi * L.k_inner + (int64_t)h * D;
    const int64_t obase = bo * L.o_outer + bi * L.o_inner + (int64_t)h * D;

    float qr[DV][4], acc[DV][4];
#pragma unroll
    for (int i = 0; i < DV; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[i][c] = 0.f; qr[i][c] = 0.f; }
    if (row_ok) {
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            ld4(q + (qbase + row * L.q_row + 16 * i + 4 * p), qr[i]);
#pragma unroll
            for (int c = 0; c < 4; ++c) qr[i][c] *= scale;
        }
    }
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < Sk; k0 += kRtKeys) {
        const int nk = min(kRtKeys, Sk - k0);
        // stage K and V rows of this head: lane l copies 4-vectors l, l+64, ... of the nk x D tile
        for (int e = lane; e < nk * (D / 4); e += 64) {
            int kr = e / (D / 4), dv = e % (D / 4);
            const T* kp = k + (kbase + (int64_t)(k0 + kr) * L.k_row + 4 * dv);
            const T* vp = v + (kbase + (int64_t)(k0 + kr) * L.k_row + 4 * dv);
            float kf[4], vf[4];
            ld4(kp, kf);
            ld4(vp, vf);
            *reinterpret_cast<float4*>(&s_k[wave][kr][4 * dv]) = make_float4(kf[0], kf[1], kf[2], kf[3]);
            *reinterpret_cast<float4*>(&s_v[wave][kr][4 * dv]) = make_float4(vf[0], vf[1], vf[2], vf[3]);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): this wave's LDS writes have landed
        float s[kRtKeys];
        float tile_max = -INFINITY;
#pragma unroll
        for (int j = 0; j < kRtKeys; ++j) {
            float d = 0.f;
            if (j < nk) {
#pragma unroll
                for (int i = 0; i < DV; ++i) {
                    const float4 kv = *reinterpret_cast<const float4*>(&s_k[wave][j][16 * i + 4 * p]);
                    d += qr[i][0] * kv.x + qr[i][1] * kv.y + qr[i][2] * kv.z + qr[i][3] * kv.w;
                }
                d += __shfl_xor(d, 1);
                d += __shfl_xor(d, 2);
                tile_max = fmaxf(tile_max, d);
            } else {
                d = -INFINITY;
            }
            s[j] = d;
        }
        const float m_new = fmaxf(m, tile_max);
        const float alpha = __expf(m - m_new);             // m = -inf on the first tile -> 0
        l *= alpha;
#pragma unroll
        for (int i = 0; i < DV; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][c] *= alpha;
#pragma unroll
        for (int j = 0; j < kRtKeys; ++j) {
            if (j < nk) {
                const float pj = __expf(s[j] - m_new);
                l += pj;
#pragma unroll
                for (int i = 0; i < DV; ++i) {
                    const float4 vv = *reinterpret_cast<const float4*>(&s_v[wave][j][16 * i + 4 * p]);
                    acc[i][0] += pj * vv.x; acc[i][1] += pj * vv.y; acc[i][2] += pj * vv.z; acc[i][3] += pj * vv.w;
                }
            }
        }
        m = m_new;
        __builtin_amdgcn_wave_barrier();
    }
    if (row_ok) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            float o4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) o4[c] = acc[i][c] * inv;
            st4(out + (obase + row * L.o_row + 16 * i + 4 * p), o4);
        }
    }
}

// q_ts / kv_ts / o_ts: elements between consecutive tokens of q, of k and v, of out (0 = H*D, the contiguous
// [.., H, D] layout). A packed projection [.., 3 H D] = (q | k | v) is addressed with q_ts = kv_ts = 3 H D and the
// three base pointers H D apart.
template <typename T>
int attn_rowtile_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk, int D,
                        float scale, hipStream_t st, int temporal_inner, int64_t q_ts, int64_t kv_ts, int64_t o_ts) {
    RtLayout L;
    const int64_t hd = (int64_t)H * D;
    if (q_ts == 0) q_ts = hd;
    if (kv_ts == 0) kv_ts = hd;
    if (o_ts == 0) o_ts = hd;
    if (temporal_inner > 0) {   // B = outer * temporal_inner problems; rows are frames strided by inner tokens
        L.inner = temporal_inner;
        L.q_inner = q_ts; L.k_inner = kv_ts; L.o_inner = o_ts;
        L.q_row = (int64_t)temporal_inner * q_ts;
        L.k_row = (int64_t)temporal_inner * kv_ts;
        L.o_row = (int64_t)temporal_inner * o_ts;
        L.q_outer = (int64_t)Sq * L.q_row;
        L.k_outer = (int64_t)Sk * L.k_row;
        L.o_outer = (int64_t)Sq * L.o_row;
    } else {
        L.inner = 1;
        L.q_inner = L.k_inner = L.o_inner = 0;
        L.q_row = q_ts; L.k_row = kv_ts; L.o_row = o_ts;
        L.q_outer = (int64_t)Sq * q_ts;
        L.k_outer = (int64_t)Sk * kv_ts;
        L.o_outer = (int64_t)Sq * o_ts;
    }
    const int q_tiles = (Sq + 15) / 16;
    const int64_t n = (int64_t)B * H * q_tiles;
    const int64_t blocks = (n + kRtWaves - 1) / kRtWaves;
    if (blocks > 0x7FFFFFFFll) return MVI_EINVAL;
    dim3 grid((unsigned)blocks), blk(64 * kRtWaves);
#define MVI_RT(DVV)                                                                                                     \
    if (Sk <= 16)                                                                                                       \
        hipLaunchKernelGGL((attn_rowtile_kernel<T, DVV, 16>), grid, blk, 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)out, \
                           B, H, Sq, Sk, scale, n, q_tiles, L);                                                         \
    else                                                                                                                \
        hipLaunchKernelGGL((attn_rowtile_kernel<T, DVV, kRtKeysMax>), grid, blk, 0, st, (const T*)q, (const T*)k, (const T*)v, \
                           (T*)out, B, H, Sq, Sk, scale, n, q_tiles, L)
    switch (D) {
        case 16: MVI_RT(1); break;
        case 32: MVI_RT(2); break;
        case 64: MVI_RT(4); break;
        default: return MVI_EINVAL;
    }
#undef MVI_RT
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

template int attn_rowtile_launch<float>(const void*, const void*, const void*, void*, int, int, int, int, int, float, hipStream_t, int, int64_t, int64_t, int64_t);
template int attn_rowtile_launch<__hip_bfloat16>(const void*, const void*, const void*, void*, int, int, int, int, int, float, hipStream_t, int, int64_t, int64_t, int64_t);
template int attn_rowtile_launch<__half>(const void*, const void*, const void*, void*, int, int, int, int, int, float, hipStream_t, int, int64_t, int64_t, int64_t);

}  // namespace mvi
