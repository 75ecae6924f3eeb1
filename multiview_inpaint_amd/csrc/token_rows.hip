// Token-row kernels of the transformer blocks for gfx950 — all HBM-bound, one pass each.
// The reference's blocks (svd_inpaint1/sgm/modules/attention.py:544-572, video_attention.py:110-141,
// :278-296) interleave residual adds, broadcast adds (the single-token cross-attention row, the frame-index
// embedding) and LayerNorms over the token-major activation t [R, C]; every one of those is a separate
// full-tensor pass in PyTorch (and PyTorch-ROCm's LayerNorm reaches < 1 TB/s at C = 320). Here:
//
//   add_layernorm : s_pre = x + h,  s = s_pre + row[r / row_div],  y = LayerNorm(s) * w + b
//                   (h, row, and the s_pre / s outputs are optional) — residual add(s) + the NEXT norm in one pass
//   add_lerp      : out = lerp(x + h, base, alpha[r / row_div])     — last residual add + AlphaBlender
//                   (diffusionmodules/util.py:312-372) in one pass
//   tokens_to_planes_add : out[n, c, p] = tok[n, p, c] + x_in[n, c, p]  — "b (h w) c -> b c h w" + the
//                   transformer's outer skip (attention.py:717-722) through an LDS tile, no strided traffic
//
// Row layout of add_layernorm: a row of C elements is covered by L = 2^k lanes x K 16-byte vectors, vector j of
// lane i at element (j * L + i) * V (V = 8 bf16/f16 or 4 fp32) — adjacent lanes read adjacent 16 B. The whole row
// lives in registers: mean, then the centred sum of squares (two passes over registers, fp32), reduced over the
// L lanes with xor shuffles. Intermediates that the unfused graph would materialise (s_pre, s) are rounded to the
// storage type exactly there, so the fused result equals the op-by-op result bit for bit.
#include <hip/hip_runtime.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"
#include "unet_io.h"

namespace mvi {

int unet_fail(int code, const char* msg);

struct AddLnArgs {
    const void *x, *h, *row;
    const float *w, *b;
    void *s_pre, *s, *y;
    int64_t R, row_div;
    int C, L, log2L;
    float eps;
};

template <typename T, int K>
__global__ __launch_bounds__(256) void add_layernorm_kernel(AddLnArgs a) {
    constexpr int V = Io<T>::kVec;
    const int L = a.L;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> a.log2L;
    const int li = threadIdx.x & (L - 1);
    if (r >= a.R) return;                       // rows are L-lane aligned: an L-group exits together (L <= 64)
    const int64_t base = r * a.C;
    const T* x = (const T*)a.x + base;
    const T* h = a.h ? (const T*)a.h + base : nullptr;
    const T* row = a.row ? (const T*)a.row + (r / a.row_div) * a.C : nullptr;
    // every load of the row is issued before anything is used: interleaved with the optional adds and stores (each behind its own
    // branch) the compiler waited for vmcnt(0) per piece — one memory latency per 16 bytes instead of one per row
    uint4 rx[K], rh[K], rr[K];
#pragma unroll
    for (int j = 0; j < K; ++j) rx[j] = *reinterpret_cast<const uint4*>(x + (j * L + li) * V);
    if (h) {
#pragma unroll
        for (int j = 0; j < K; ++j) rh[j] = *reinterpret_cast<const uint4*>(h + (j * L + li) * V);
    }
    if (row) {
#pragma unroll
        for (int j = 0; j < K; ++j) rr[j] = *reinterpret_cast<const uint4*>(row + (j * L + li) * V);
    }
    float v[K][V];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int e = (j * L + li) * V;
        Io<T>::load(reinterpret_cast<const T*>(&rx[j]), v[j]);
        if (h) {
            float t[V];
            Io<T>::load(reinterpret_cast<const T*>(&rh[j]), t);
#pragma unroll
            for (int k = 0; k < V; ++k) v[j][k] = round_to<T>(v[j][k] + t[k]);
        }
        if (a.s_pre) Io<T>::store((T*)a.s_pre + base + e, v[j]);
        if (row) {
            float t[V];
            Io<T>::load(reinterpret_cast<const T*>(&rr[j]), t);
#pragma unroll
            for (int k = 0; k < V; ++k) v[j][k] = round_to<T>(v[j][k] + t[k]);
        }
        if (a.s) Io<T>::store((T*)a.s + base + e, v[j]);
#pragma unroll
        for (int k = 0; k < V; ++k) sum += v[j][k];
    }
    for (int o = 1; o < L; o <<= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)a.C;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[j][k] - mean; m2 += d * d; }
    for (int o = 1; o < L; o <<= 1) m2 += __shfl_xor(m2, o);
    const float rstd = rsqrtf(m2 / (float)a.C + a.eps);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int e = (j * L + li) * V;
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = (v[j][k] - mean) * rstd * a.w[e + k] + a.b[e + k];
        Io<T>::store((T*)a.y + base + e, o);
    }
}

template <typename T>
static int add_layernorm_launch(AddLnArgs a, hipStream_t st) {
    constexpr int V = Io<T>::kVec;
    if (a.C % V != 0) return MVI_EINVAL;
    const int vecs = a.C / V;
    int L = 1, log2L = 0;
    while (L <= 64 && (vecs % L != 0 || vecs / L > 8)) { L <<= 1; ++log2L; }
    if (L > 64) return MVI_EINVAL;
    a.L = L; a.log2L = log2L;
    const int K = vecs / L;
    const int64_t threads = a.R * L;
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return MVI_EINVAL;
#define MVI_LN(KK) case KK: hipLaunchKernelGGL((add_layernorm_kernel<T, KK>), dim3((unsigned)blocks), dim3(256), 0, st, a); break
    switch (K) {
        MVI_LN(1); MVI_LN(2); MVI_LN(3); MVI_LN(4); MVI_LN(5); MVI_LN(6); MVI_LN(7); MVI_LN(8);
        default: return MVI_EINVAL;
    }
#undef MVI_LN
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// out = lerp(t, base, alpha) with t = x + h, in the two-sided form PyTorch's lerp uses
// (weight < 0.5 ? start + w (end - start) : end - (end - start)(1 - w))
template <typename T>
__global__ __launch_bounds__(256) void add_lerp_kernel(const T* __restrict__ x, const T* __restrict__ h,
                                                       const T* __restrict__ base, const float* __restrict__ alpha,
                                                       T* __restrict__ out, int64_t n_vec, int vec_per_row, int64_t row_div) {
    constexpr int V = Io<T>::kVec;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    const int64_t r = i / vec_per_row;
    const float w = alpha[r / row_div];
    float a[V], b[V], c[V];
    Io<T>::load(x + i * V, a);
    Io<T>::load(base + i * V, c);
    if (h) {
        Io<T>::load(h + i * V, b);
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] = round_to<T>(a[k] + b[k]);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float d = c[k] - a[k];
        a[k] = w < 0.5f ? a[k] + w * d : c[k] - d * (1.0f - w);
    }
    Io<T>::store(out + i * V, a);
}

// out[n, c, p] = tok[n, p, c] + x_in[n, c, p]; 64 x 64 (p x c) tiles through LDS
constexpr int kTpTile = 64;
// base (optional, tokens like tok) + alpha [N] fp32: out = base + (1 - alpha[n]) * (tok + bias) — the tail of VideoResBlock on tokens:
// the temporal ResBlock's skip add and the AlphaBlender (video_model.py:67-81, util.py:358-372) in the pass that restores b c h w.
template <typename T, bool kBlend>
__global__ __launch_bounds__(256) void tokens_to_planes_add_kernel(const T* __restrict__ tok, const T* __restrict__ x_in,
                                                                   T* __restrict__ out, int C, int64_t S, int p_tiles,
                                                                   int c_tiles, const float* __restrict__ bias,
                                                                   const T* __restrict__ base = nullptr, const float* __restrict__ alpha = nullptr) {
    constexpr int V = Io<T>::kVec;
    constexpr int VPR = kTpTile / V;            // 16-B vectors per tile row
    __shared__ float s_t[kTpTile][kTpTile + 1];
    __shared__ float s_b[kBlend ? kTpTile : 1][kTpTile + 1];  // (base != nullptr exactly when kBlend)
    int bid = blockIdx.x;
    const int pt = bid % p_tiles; bid /= p_tiles;
    const int ct = bid % c_tiles;
    const int64_t n = bid / c_tiles;
    const int64_t p0 = (int64_t)pt * kTpTile;
    const int c0 = ct * kTpTile;
    // Every load of a phase is issued before the first use, from an address clamped into the tensor instead of behind its bounds
    // test: a load inside a branch makes the compiler wait vmcnt(0) at the first use — one memory latency per piece.
    constexpr int kIt = kTpTile * VPR / 256;     // pieces per thread and phase (2 in bf16 / f16, 4 in fp32)
    // read tok[n, p0 + pr, c0 + 8 cv ..]: rows of the token-major tile
    uint4 rt[kIt], rb[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = threadIdx.x + 256 * it, pr = i / VPR, cv = i % VPR;
        const bool ok = p0 + pr < S && c0 + cv * V < C;
        const int64_t o = ok ? (n * S + p0 + pr) * C + c0 + cv * V : n * S * C;
        rt[it] = *reinterpret_cast<const uint4*>(tok + o);
        if (kBlend) rb[it] = *reinterpret_cast<const uint4*>(base + o);
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = threadIdx.x + 256 * it, pr = i / VPR, cv = i % VPR;
        if (p0 + pr < S && c0 + cv * V < C) {
            float t[V];
            Io<T>::load(reinterpret_cast<const T*>(&rt[it]), t);
#pragma unroll
            for (int k = 0; k < V; ++k) s_t[pr][cv * V + k] = t[k];
            if (kBlend) {
                Io<T>::load(reinterpret_cast<const T*>(&rb[it]), t);
#pragma unroll
                for (int k = 0; k < V; ++k) s_b[pr][cv * V + k] = t[k];
            }
        }
    }
    const float wgt = kBlend ? 1.0f - alpha[n] : 1.0f;
    // write out[n, c0 + cr, p0 + 8 pv ..]
    uint4 rx[kIt];
    if (!kBlend && x_in) {
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = threadIdx.x + 256 * it, cr = i / VPR, pv = i % VPR;
            const bool ok = c0 + cr < C && p0 + pv * V < S;
            rx[it] = *reinterpret_cast<const uint4*>(x_in + (ok ? (n * C + c0 + cr) * S + p0 + pv * V : n * C * S));
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = threadIdx.x + 256 * it, cr = i / VPR, pv = i % VPR;
        if (c0 + cr < C && p0 + pv * V < S) {
            const int64_t o = (n * C + c0 + cr) * S + p0 + pv * V;
            float t[V], xi[V];
            if (!kBlend && x_in) {
                Io<T>::load(reinterpret_cast<const T*>(&rx[it]), xi);
            } else {                                              // plain "b (h w) c -> b c h w"
#pragma unroll
                for (int k = 0; k < V; ++k) xi[k] = 0.f;
            }
            const float bc = bias ? bias[c0 + cr] : 0.f;          // per-channel bias of the convolution that produced the tokens
#pragma unroll
            for (int k = 0; k < V; ++k) t[k] = kBlend ? s_b[pv * V + k][cr] + wgt * (s_t[pv * V + k][cr] + bc) : s_t[pv * V + k][cr] + bc + xi[k];
            Io<T>::store(out + o, t);
        }
    }
}


// out[n, up(p), c] = x[n, c, p]: "b c h w -> b (h w) c" through the same 64 x 64 LDS tile, optionally with the nearest-neighbour 2x
// upsampling of Upsample (openaimodel.py:118-134, F.interpolate(scale_factor=2, mode="nearest")) folded into the token write: source
// pixel (y, x) of a W-wide image lands on tokens (2 y + a)(2 W) + 2 x + b, a, b in {0, 1} — the 4x larger NCHW tensor never exists.
// tok_add (optional, tokens) / bias (optional, fp32 [C]): out = x^T + tok_add + bias — the last add of a ResBlock whose result stays
// token-major for the temporal ResBlock that follows (openaimodel.py:354 `skip_connection(x) + h` with h and the result as tokens).
template <typename T>
__global__ __launch_bounds__(256) void planes_to_tokens_kernel(const T* __restrict__ x, T* __restrict__ out, int C, int64_t S, int p_tiles,
                                                               int c_tiles, int W, int up, const T* __restrict__ tok_add = nullptr,
                                                               const float* __restrict__ bias = nullptr) {
    constexpr int V = Io<T>::kVec;
    constexpr int VPR = kTpTile / V;
    __shared__ float s_t[kTpTile][kTpTile + 1];      // [c][p]
    int bid = blockIdx.x;
    const int pt = bid % p_tiles; bid /= p_tiles;
    const int ct = bid % c_tiles;
    const int64_t n = bid / c_tiles;
    const int64_t p0 = (int64_t)pt * kTpTile;
    const int c0 = ct * kTpTile;
    constexpr int kIt = kTpTile * VPR / 256;     // (loads hoisted and clamped as in tokens_to_planes_add_kernel)
    uint4 rx[kIt], ra[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = threadIdx.x + 256 * it, cr = i / VPR, pv = i % VPR;
        const bool ok = c0 + cr < C && p0 + pv * V < S;
        rx[it] = *reinterpret_cast<const uint4*>(x + (ok ? (n * C + c0 + cr) * S + p0 + pv * V : n * C * S));
    }
    const int64_t So = up == 2 ? 4 * S : S;
    float bb[kIt][V];                                // the bias of the thread's channels (phase 2): loaded here, not per element there
    if (tok_add) {                                   // (up == 1)
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = threadIdx.x + 256 * it, pr = i / VPR, cv = i % VPR;
            const bool ok = p0 + pr < S && c0 + cv * V < C;
            ra[it] = *reinterpret_cast<const uint4*>(tok_add + (ok ? (n * So + p0 + pr) * C + c0 + cv * V : n * So * C));
#pragma unroll
            for (int k = 0; k < V; ++k) bb[it][k] = bias ? bias[ok ? c0 + cv * V + k : 0] : 0.f;
        }
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = threadIdx.x + 256 * it, cr = i / VPR, pv = i % VPR;
        if (c0 + cr < C && p0 + pv * V < S) {
            float t[V];
            Io<T>::load(reinterpret_cast<const T*>(&rx[it]), t);
#pragma unroll
            for (int k = 0; k < V; ++k) s_t[cr][pv * V + k] = t[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = threadIdx.x + 256 * it, pr = i / VPR, cv = i % VPR;
        const int64_t p = p0 + pr;
        if (p < S && c0 + cv * V < C) {
            float t[V];
#pragma unroll
            for (int k = 0; k < V; ++k) t[k] = s_t[cv * V + k][pr];
            if (up == 2) {
                const int64_t y = p / W, xx = p - y * W;
                T* const o = out + ((n * So + (2 * y) * (2 * (int64_t)W) + 2 * xx) * C + c0 + cv * V);
                Io<T>::store(o, t);
                Io<T>::store(o + C, t);
                Io<T>::store(o + 2 * (int64_t)W * C, t);
                Io<T>::store(o + (2 * (int64_t)W + 1) * C, t);
            } else {
                const int64_t o = (n * So + p) * C + c0 + cv * V;
                if (tok_add) {
                    float a[V];
                    Io<T>::load(reinterpret_cast<const T*>(&ra[it]), a);
#pragma unroll
                    for (int k = 0; k < V; ++k) t[k] += a[k] + bb[it][k];
                }
                Io<T>::store(out + o, t);
            }
        }
    }
}

}  // namespace mvi

using namespace mvi;

extern "C" int mvi_add_layernorm(const void* x, const void* h, const void* row, int64_t row_div, const float* weight,
                                 const float* bias, void* s_pre, void* s, void* y, int64_t R, int32_t C, float eps,
                                 int32_t dtype, void* stream) {
    if (R < 0 || C <= 0) return unet_fail(MVI_EINVAL, "add_layernorm: bad shape");
    if (R == 0) return MVI_OK;
    if (!x || !weight || !bias || !y) return unet_fail(MVI_EINVAL, "add_layernorm: NULL pointer");
    if (row && row_div <= 0) return unet_fail(MVI_EINVAL, "add_layernorm: row_div must be positive");
    if (s_pre && !h) return unet_fail(MVI_EINVAL, "add_layernorm: s_pre requested without h");
    if (s && !h && !row) return unet_fail(MVI_EINVAL, "add_layernorm: s requested without h or row");
    AddLnArgs a{x, h, row, weight, bias, s_pre, s, y, R, row ? row_div : 1, C, 0, 0, eps};
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = add_layernorm_launch<float>(a, (hipStream_t)stream); break;
        case MVI_DT_BF16: rc = add_layernorm_launch<__hip_bfloat16>(a, (hipStream_t)stream); break;
        case MVI_DT_F16: rc = add_layernorm_launch<__half>(a, (hipStream_t)stream); break;
        default: return unet_fail(MVI_EINVAL, "add_layernorm: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "add_layernorm: C must split into 2^k lanes x <= 8 16-byte vectors");
    return rc ? unet_fail(MVI_EHIP, "add_layernorm: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_layernorm_supported(int32_t C, int32_t dtype) {
    const int V = dtype == MVI_DT_F32 ? 4 : 8;
    if (C <= 0 || C % V) return 0;
    const int vecs = C / V;
    for (int L = 1; L <= 64; L <<= 1)
        if (vecs % L == 0 && vecs / L <= 8) return 1;
    return 0;
}

template <typename T>
static int add_lerp_launch(const void* x, const void* h, const void* base, const float* alpha, int64_t row_div, void* out,
                           int64_t R, int C, hipStream_t st) {
    constexpr int V = Io<T>::kVec;
    if (C % V) return MVI_EINVAL;
    const int64_t n_vec = R * (C / V);
    const int64_t blocks = (n_vec + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return MVI_EINVAL;
    hipLaunchKernelGGL((add_lerp_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (const T*)h,
                       (const T*)base, alpha, (T*)out, n_vec, C / V, row_div);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

extern "C" int mvi_add_lerp(const void* x, const void* h, const void* base, const float* alpha, int64_t row_div, void* out,
                            int64_t R, int32_t C, int32_t dtype, void* stream) {
    if (R < 0 || C <= 0 || row_div <= 0) return unet_fail(MVI_EINVAL, "add_lerp: bad shape");
    if (R == 0) return MVI_OK;
    if (!x || !base || !alpha || !out) return unet_fail(MVI_EINVAL, "add_lerp: NULL pointer");
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = add_lerp_launch<float>(x, h, base, alpha, row_div, out, R, C, (hipStream_t)stream); break;
        case MVI_DT_BF16: rc = add_lerp_launch<__hip_bfloat16>(x, h, base, alpha, row_div, out, R, C, (hipStream_t)stream); break;
        case MVI_DT_F16: rc = add_lerp_launch<__half>(x, h, base, alpha, row_div, out, R, C, (hipStream_t)stream); break;
        default: return unet_fail(MVI_EINVAL, "add_lerp: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "add_lerp: C must be a multiple of the 16-byte vector width");
    return rc ? unet_fail(MVI_EHIP, "add_lerp: kernel launch failed") : MVI_OK;
}

// out[r][c] = a[r][c] + alpha * (b[r][c] + bias[c]) on fp32 rows [R, C] (round 6): the two adds that end a ResBlock of the first-stage
// decoder on token-major fp32 activations (svd/vae_split.py) — skip + convolution output + its bias, and the AlphaBlender form
// x + alpha (conv + bias) of temporal_ae.py:70-81 — as ONE pass (a, b read, out written) instead of two in-place PyTorch adds each.
// out may alias a or b. bias may be NULL.
__global__ __launch_bounds__(256) void rows_axpb_kernel(const float* a, const float* b, const float* __restrict__ bias, float* out, float alpha,
                                                        int64_t n4, int c4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 av = reinterpret_cast<const float4*>(a)[i], bv = reinterpret_cast<const float4*>(b)[i];
    float4 cv = {0.f, 0.f, 0.f, 0.f};
    if (bias) cv = reinterpret_cast<const float4*>(bias)[i % c4];
    float4 o;
    o.x = av.x + alpha * (bv.x + cv.x); o.y = av.y + alpha * (bv.y + cv.y); o.z = av.z + alpha * (bv.z + cv.z); o.w = av.w + alpha * (bv.w + cv.w);
    reinterpret_cast<float4*>(out)[i] = o;
}

extern "C" int mvi_rows_axpb_f32(const float* a, const float* b, const float* bias, float alpha, float* out, int64_t R, int32_t C, void* stream) {
    if (R < 0 || C <= 0 || C % 4) return unet_fail(MVI_EINVAL, "rows_axpb_f32: C must be a positive multiple of 4");
    if (R == 0) return MVI_OK;
    if (!a || !b || !out) return unet_fail(MVI_EINVAL, "rows_axpb_f32: NULL pointer");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)bias) % 16) return unet_fail(MVI_EINVAL, "rows_axpb_f32: pointers must be 16-byte aligned");
    const int64_t n4 = R * (C / 4), blocks = (n4 + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return unet_fail(MVI_EINVAL, "rows_axpb_f32: too many rows for one launch");
    hipLaunchKernelGGL(rows_axpb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, bias, out, alpha, n4, C / 4);
    return hipGetLastError() == hipSuccess ? MVI_OK : unet_fail(MVI_EHIP, "rows_axpb_f32: kernel launch failed");
}

template <typename T>
static int tokens_to_planes_launch(const void* tok, const void* x_in, void* out, int64_t N, int C, int64_t S, hipStream_t st,
                                   const float* bias = nullptr, const void* base = nullptr, const float* alpha = nullptr) {
    constexpr int V = Io<T>::kVec;
    if (C % V || S % V) return MVI_EINVAL;
    const int p_tiles = (int)((S + kTpTile - 1) / kTpTile), c_tiles = (C + kTpTile - 1) / kTpTile;
    const int64_t blocks = N * p_tiles * c_tiles;
    if (blocks > 0x7FFFFFFFll) return MVI_EINVAL;
    if (base)
        hipLaunchKernelGGL((tokens_to_planes_add_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)tok,
                           (const T*)x_in, (T*)out, C, S, p_tiles, c_tiles, bias, (const T*)base, alpha);
    else
        hipLaunchKernelGGL((tokens_to_planes_add_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)tok,
                           (const T*)x_in, (T*)out, C, S, p_tiles, c_tiles, bias, (const T*)base, alpha);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

extern "C" int mvi_tokens_to_planes_add(const void* tok, const void* x_in, void* out, int64_t N, int32_t C, int64_t spatial,
                                        int32_t dtype, void* stream) {
    if (N < 0 || C <= 0 || spatial < 0) return unet_fail(MVI_EINVAL, "tokens_to_planes_add: bad shape");
    if (N == 0 || spatial == 0) return MVI_OK;
    if (!tok || !out) return unet_fail(MVI_EINVAL, "tokens_to_planes_add: NULL pointer");
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = tokens_to_planes_launch<float>(tok, x_in, out, N, C, spatial, (hipStream_t)stream); break;
        case MVI_DT_BF16: rc = tokens_to_planes_launch<__hip_bfloat16>(tok, x_in, out, N, C, spatial, (hipStream_t)stream); break;
        case MVI_DT_F16: rc = tokens_to_planes_launch<__half>(tok, x_in, out, N, C, spatial, (hipStream_t)stream); break;
        default: return unet_fail(MVI_EINVAL, "tokens_to_planes_add: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "tokens_to_planes_add: C and spatial must be multiples of the 16-byte vector width");
    return rc ? unet_fail(MVI_EHIP, "tokens_to_planes_add: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_tokens_to_planes_add_bias(const void* tok, const void* x_in, const float* bias, void* out, int64_t N, int32_t C, int64_t spatial,
                                        int32_t dtype, void* stream) {
    if (N < 0 || C <= 0 || spatial < 0) return unet_fail(MVI_EINVAL, "tokens_to_planes_add_bias: bad shape");
    if (N == 0 || spatial == 0) return MVI_OK;
    if (!tok || !out) return unet_fail(MVI_EINVAL, "tokens_to_planes_add_bias: NULL pointer");
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = tokens_to_planes_launch<float>(tok, x_in, out, N, C, spatial, (hipStream_t)stream, bias); break;
        case MVI_DT_BF16: rc = tokens_to_planes_launch<__hip_bfloat16>(tok, x_in, out, N, C, spatial, (hipStream_t)stream, bias); break;
        case MVI_DT_F16: rc = tokens_to_planes_launch<__half>(tok, x_in, out, N, C, spatial, (hipStream_t)stream, bias); break;
        default: return unet_fail(MVI_EINVAL, "tokens_to_planes_add_bias: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "tokens_to_planes_add_bias: C and spatial must be multiples of the 16-byte vector width");
    return rc ? unet_fail(MVI_EHIP, "tokens_to_planes_add_bias: kernel launch failed") : MVI_OK;
}

template <typename T>
static int planes_to_tokens_launch(const void* x, void* out, int64_t N, int C, int64_t S, int W, int up, hipStream_t st,
                                   const void* tok_add = nullptr, const float* bias = nullptr) {
    constexpr int V = Io<T>::kVec;
    if (C % V || S % V) return MVI_EINVAL;
    const int p_tiles = (int)((S + kTpTile - 1) / kTpTile), c_tiles = (C + kTpTile - 1) / kTpTile;
    const int64_t blocks = N * p_tiles * c_tiles;
    if (blocks > 0x7FFFFFFFll) return MVI_EINVAL;
    hipLaunchKernelGGL((planes_to_tokens_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)out, C, S, p_tiles, c_tiles, W, up,
                       (const T*)tok_add, bias);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

extern "C" int mvi_planes_to_tokens(const void* x, void* out, int64_t N, int32_t C, int32_t H, int32_t W, int32_t upsample, int32_t dtype,
                                    void* stream) {
    if (N < 0 || C <= 0 || H < 0 || W < 0 || (upsample != 1 && upsample != 2)) return unet_fail(MVI_EINVAL, "planes_to_tokens: bad shape / upsample must be 1 or 2");
    const int64_t S = (int64_t)H * W;
    if (N == 0 || S == 0) return MVI_OK;
    if (!x || !out) return unet_fail(MVI_EINVAL, "planes_to_tokens: NULL pointer");
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = planes_to_tokens_launch<float>(x, out, N, C, S, W, upsample, (hipStream_t)stream); break;
        case MVI_DT_BF16: rc = planes_to_tokens_launch<__hip_bfloat16>(x, out, N, C, S, W, upsample, (hipStream_t)stream); break;
        case MVI_DT_F16: rc = planes_to_tokens_launch<__half>(x, out, N, C, S, W, upsample, (hipStream_t)stream); break;
        default: return unet_fail(MVI_EINVAL, "planes_to_tokens: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "planes_to_tokens: C and H W must be multiples of the 16-byte vector width");
    return rc ? unet_fail(MVI_EHIP, "planes_to_tokens: kernel launch failed") : MVI_OK;
}

/* out[n, p, c] = x[n, c, p] + tok[n, p, c] + bias[c] (bias optional): a ResBlock's last add with tokens in and tokens out. */
extern "C" int mvi_planes_add_to_tokens(const void* x, const void* tok, const float* bias, void* out, int64_t N, int32_t C, int64_t spatial,
                                        int32_t dtype, void* stream) {
    if (N < 0 || C <= 0 || spatial < 0) return unet_fail(MVI_EINVAL, "planes_add_to_tokens: bad shape");
    if (N == 0 || spatial == 0) return MVI_OK;
    if (!x || !tok || !out) return unet_fail(MVI_EINVAL, "planes_add_to_tokens: NULL pointer");
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = planes_to_tokens_launch<float>(x, out, N, C, spatial, 1, 1, (hipStream_t)stream, tok, bias); break;
        case MVI_DT_BF16: rc = planes_to_tokens_launch<__hip_bfloat16>(x, out, N, C, spatial, 1, 1, (hipStream_t)stream, tok, bias); break;
        case MVI_DT_F16: rc = planes_to_tokens_launch<__half>(x, out, N, C, spatial, 1, 1, (hipStream_t)stream, tok, bias); break;
        default: return unet_fail(MVI_EINVAL, "planes_add_to_tokens: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "planes_add_to_tokens: C and spatial must be multiples of the 16-byte vector width");
    return rc ? unet_fail(MVI_EHIP, "planes_add_to_tokens: kernel launch failed") : MVI_OK;
}

/* out[n, c, p] = base[n, p, c] + (1 - alpha[n]) * (tok[n, p, c] + bias[c]): the tail of a VideoResBlock evaluated on tokens. */
extern "C" int mvi_tokens_blend_to_planes(const void* tok, const void* base, const float* bias, const float* alpha, void* out, int64_t N,
                                          int32_t C, int64_t spatial, int32_t dtype, void* stream) {
    if (N < 0 || C <= 0 || spatial < 0) return unet_fail(MVI_EINVAL, "tokens_blend_to_planes: bad shape");
    if (N == 0 || spatial == 0) return MVI_OK;
    if (!tok || !base || !alpha || !out) return unet_fail(MVI_EINVAL, "tokens_blend_to_planes: NULL pointer");
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = tokens_to_planes_launch<float>(tok, nullptr, out, N, C, spatial, (hipStream_t)stream, bias, base, alpha); break;
        case MVI_DT_BF16: rc = tokens_to_planes_launch<__hip_bfloat16>(tok, nullptr, out, N, C, spatial, (hipStream_t)stream, bias, base, alpha); break;
        case MVI_DT_F16: rc = tokens_to_planes_launch<__half>(tok, nullptr, out, N, C, spatial, (hipStream_t)stream, bias, base, alpha); break;
        default: return unet_fail(MVI_EINVAL, "tokens_blend_to_planes: unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "tokens_blend_to_planes: C and spatial must be multiples of the 16-byte vector width");
    return rc ? unet_fail(MVI_EHIP, "tokens_blend_to_planes: kernel launch failed") : MVI_OK;
}
