// Fused GroupNorm(+SiLU) on channels-last activations for gfx950 — HBM-bound: 2 reads + 1 write of the tensor.
// Same op as groupnorm_silu.hip (svd_inpaint1/sgm/modules/diffusionmodules/openaimodel.py:257-261, :292-305; util.py:274-276;
// temporal form video_model.py:71-75) for x laid out [(videos T), S, C] with the channels innermost — the memory of a
// torch.channels_last [N, C, H, W] tensor, and at the same time the token-major [N, (h w), C] tensor of the transformer
// stems. With the UNet held in this layout MIOpen / CK run their NHWC convolution kernels directly (measured: the 3x3
// convolution at [28, 320, 72, 128] 0.98 -> 0.69 ms, no batched_transpose kernels around it), the 1x1 convolutions and
// the (3,1,1) temporal convolutions are plain GEMMs over the token rows, and "b c h w <-> b (h w) c" costs nothing.
//
// A group is Cg adjacent channels of every row of a video (T frames x S positions). Three launches:
//   stats    : grid (chunks of 256 rows, videos). Thread (slot, column) owns KT 16-byte channel vectors of every RS-th
//              row; it accumulates SHIFTED sums (shift = the first value it sees per channel, so the variance does
//              not cancel), the block turns them into per-channel (n, mean, M2) in LDS and one thread per group
//              merges the group's channels and row slots with Chan's formula -> partial (n, mean, M2);
//   finalize : one wave per (video, group) merges the chunks' partials (64 lanes + butterfly) -> mean, rstd;
//   apply    : same grid as stats; per-channel scale / shift in LDS, y = act(a (x + chan_bias) + b), written in the
//              same layout or channel-stacked for the temporal convolution (stack3: rows of 3C = frame t-1 | t | t+1).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"
#include "unet_io.h"

namespace mvi {

int unet_fail(int code, const char* msg);

constexpr int kNhBlock = 256;
constexpr int kNhRows = 256;           // most rows per block; small tensors take fewer so that ~1000 blocks exist
constexpr int kNhUnroll = 4;           // 16-byte loads a thread keeps in flight

struct NhGeom {
    int64_t R;       // rows per video = T * S
    int64_t S;       // positions per frame
    int C, G, Cg, T;
    int TPR;         // threads per row = (C / vec) / KT
    int RS;          // row slots per block = 256 / TPR
    int cpv;         // chunks per video
    int rpb;         // rows per block (<= kNhRows)
    const float* chan_bias;   // optional [(videos T), C], added to x before the statistics
    int stack3;
};

__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    const float nt = n + nb;
    if (nb > 0.f) {
        const float d = mb - mean;
        mean += d * (nb / nt);
        m2 += m2b + d * d * (n * nb / nt);
        n = nt;
    }
}

template <typename T, int KT>
__global__ __launch_bounds__(kNhBlock) void gn_nhwc_stats_kernel(const T* __restrict__ x, float* __restrict__ part, NhGeom q) {
    constexpr int V = Io<T>::kVec;
    extern __shared__ float s_dyn[];
    const int C = q.C, RS = q.RS, TPR = q.TPR;
    float* s_mean = s_dyn;
    float* s_m2 = s_dyn + (size_t)RS * C;
    float* s_n = s_m2 + (size_t)RS * C;
    const int tid = threadIdx.x;
    const int slot = tid / TPR, tv = tid - slot * TPR;
    const int64_t video = blockIdx.y, row0 = (int64_t)blockIdx.x * q.rpb;
    const int rows = (int)((q.R - row0) < q.rpb ? (q.R - row0) : q.rpb);
    const T* xb = x + (video * q.R + row0) * C;
    float K[KT][V], s1[KT][V], s2[KT][V];
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int k = 0; k < V; ++k) { K[j][k] = 0.f; s1[j][k] = 0.f; s2[j][k] = 0.f; }
    int n = 0;
    if (slot < RS) {
        // rows r = slot, slot + RS, ...; kNhUnroll rows are requested before any of them is consumed (one load in
        // flight per thread left the kernel latency-bound). Frame index by 32-bit counting, no 64-bit division per row.
        constexpr int U = kNhUnroll / KT > 0 ? kNhUnroll / KT : 1;
        const int S32 = (int)q.S;
        int frame = (int)(row0 / q.S), fs = (int)(row0 - (int64_t)frame * q.S) + slot;     // position of row `slot`
        while (fs >= S32) { fs -= S32; ++frame; }
        for (int r = slot; r < rows; r += U * RS) {
            float v[U][KT][V];
            int fr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ru = r + u * RS;
                fr[u] = frame;
                fs += RS;
                while (fs >= S32) { fs -= S32; ++frame; }
                if (ru < rows) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) Io<T>::load(xb + (int64_t)ru * C + (tv + j * TPR) * V, v[u][j]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u * RS >= rows) break;
                const float* cb = q.chan_bias ? q.chan_bias + (video * q.T + fr[u]) * C : nullptr;
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    const int c0 = (tv + j * TPR) * V;
                    if (cb) {
#pragma unroll
                        for (int k = 0; k < V; k += 4) {
                            const float4 cv = *reinterpret_cast<const float4*>(cb + c0 + k);
                            v[u][j][k] += cv.x; v[u][j][k + 1] += cv.y; v[u][j][k + 2] += cv.z; v[u][j][k + 3] += cv.w;
                        }
                    }
                    if (n == 0) {
#pragma unroll
                        for (int k = 0; k < V; ++k) K[j][k] = v[u][j][k];
                    }
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const float d = v[u][j][k] - K[j][k];
                        s1[j][k] += d;
                        s2[j][k] = __builtin_fmaf(d, d, s2[j][k]);
                    }
                }
                ++n;
            }
        }
        const float fn = (float)n, inv = n > 0 ? 1.0f / fn : 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int c = (tv + j * TPR) * V + k;
                const float ds = s1[j][k] * inv;
                s_mean[(size_t)slot * C + c] = K[j][k] + ds;
                s_m2[(size_t)slot * C + c] = fmaxf(s2[j][k] - s1[j][k] * ds, 0.f);
            }
        if (tv == 0) s_n[slot] = fn;
    }
    __syncthreads();
    for (int g = tid; g < q.G; g += kNhBlock) {
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
        for (int sl = 0; sl < RS; ++sl) {
            const float ns = s_n[sl];
            if (ns <= 0.f) continue;
            for (int cc = 0; cc < q.Cg; ++cc) {
                const int c = g * q.Cg + cc;
                chan_merge(cnt, mean, m2, ns, s_mean[(size_t)sl * C + c], s_m2[(size_t)sl * C + c]);
            }
        }
        float* p = part + ((video * q.cpv + blockIdx.x) * q.G + g) * 3;
        p[0] = cnt; p[1] = mean; p[2] = m2;
    }
}

// grid (videos, ceil(G / 4)): one wave per group walks the chunks' partials (64 lanes), butterfly, writes (mean, rstd)
__global__ __launch_bounds__(kNhBlock) void gn_nhwc_finalize_kernel(const float* __restrict__ part, float* __restrict__ stat,
                                                                    int cpv, int G, float eps) {
    const int64_t video = blockIdx.x;
    const int g = blockIdx.y * (kNhBlock / 64) + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (g >= G) return;                                                  // whole wave
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    for (int c = l; c < cpv; c += 64) {
        const float* p = part + ((video * cpv + c) * G + g) * 3;
        chan_merge(cnt, mean, m2, p[0], p[1], p[2]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float nb = __shfl_xor(cnt, o), mb = __shfl_xor(mean, o), m2b = __shfl_xor(m2, o);
        chan_merge(cnt, mean, m2, nb, mb, m2b);
    }
    if (l == 0) {
        stat[(video * G + g) * 2] = mean;
        stat[(video * G + g) * 2 + 1] = rsqrtf(m2 / cnt + eps);
    }
}

template <typename T, int KT>
__global__ __launch_bounds__(kNhBlock) void gn_nhwc_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                 const float* __restrict__ weight,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ stat, NhGeom q, int silu) {
    constexpr int V = Io<T>::kVec;
    extern __shared__ float s_dyn[];
    const int C = q.C, RS = q.RS, TPR = q.TPR;
    float* s_a = s_dyn;
    float* s_b = s_dyn + C;
    const int tid = threadIdx.x;
    const int64_t video = blockIdx.y, row0 = (int64_t)blockIdx.x * q.rpb;
    for (int c = tid; c < C; c += kNhBlock) {
        const int g = c / q.Cg;
        const float mean = stat[(video * q.G + g) * 2], rstd = stat[(video * q.G + g) * 2 + 1];
        const float a = weight[c] * rstd;
        s_a[c] = a;
        s_b[c] = bias[c] - mean * a;
    }
    __syncthreads();
    const int slot = tid / TPR, tv = tid - slot * TPR;
    if (slot >= RS) return;
    const int rows = (int)((q.R - row0) < q.rpb ? (q.R - row0) : q.rpb);
    float a[KT][V], b[KT][V];
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = (tv + j * TPR) * V + k;
            a[j][k] = s_a[c];
            b[j][k] = s_b[c];
        }
    const T* xb = x + (video * q.R + row0) * C;
    constexpr int U = kNhUnroll / KT > 0 ? kNhUnroll / KT : 1;
    const int S32 = (int)q.S;
    int frame = (int)(row0 / q.S), fs = (int)(row0 - (int64_t)frame * q.S) + slot;
    while (fs >= S32) { fs -= S32; ++frame; }
    const int64_t C3 = 3 * (int64_t)C;
    for (int r = slot; r < rows; r += U * RS) {
        float v[U][KT][V];
        int fr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ru = r + u * RS;
            fr[u] = frame;
            fs += RS;
            while (fs >= S32) { fs -= S32; ++frame; }
            if (ru < rows) {
#pragma unroll
                for (int j = 0; j < KT; ++j) Io<T>::load(xb + (int64_t)ru * C + (tv + j * TPR) * V, v[u][j]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ru = r + u * RS;
            if (ru >= rows) break;
            const float* cb = q.chan_bias ? q.chan_bias + (video * q.T + fr[u]) * C : nullptr;
            const int64_t grow = video * q.R + row0 + ru;             // global row
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int c0 = (tv + j * TPR) * V;
                float* w = v[u][j];
                if (cb) {
#pragma unroll
                    for (int k = 0; k < V; k += 4) {
                        const float4 cv = *reinterpret_cast<const float4*>(cb + c0 + k);
                        w[k] += cv.x; w[k + 1] += cv.y; w[k + 2] += cv.z; w[k + 3] += cv.w;
                    }
                }
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float t = __builtin_fmaf(w[k], a[j][k], b[j][k]);
                    w[k] = silu ? t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)) : t;
                }
                if (!q.stack3) {
                    Io<T>::store(y + grow * C + c0, w);
                } else {
                    Io<T>::store(y + grow * C3 + C + c0, w);                                          // tap 1 of frame t
                    if (fr[u] + 1 < q.T) Io<T>::store(y + (grow + q.S) * C3 + c0, w);                  // tap 0 of frame t+1
                    if (fr[u] > 0) Io<T>::store(y + (grow - q.S) * C3 + 2 * (int64_t)C + c0, w);       // tap 2 of frame t-1
                    float z[V];
#pragma unroll
                    for (int k = 0; k < V; ++k) z[k] = 0.f;
                    if (fr[u] == 0) Io<T>::store(y + grow * C3 + c0, z);                              // zero frame before the first
                    if (fr[u] == q.T - 1) Io<T>::store(y + grow * C3 + 2 * (int64_t)C + c0, z);       // ... and after the last
                }
            }
        }
    }
}

// rows per block: kNhRows for large tensors; fewer (a multiple of the row slots, at least 4 rows per slot) when the
// tensor is small, so that about a thousand blocks exist and no thread walks hundreds of rows on its own
static int nh_rows_per_block(int64_t videos, int64_t R, int RS) {
    int64_t want = (videos * R + 1023) / 1024;
    const int64_t lo = 4 * (int64_t)RS;
    if (want < lo) want = lo;
    want = (want + RS - 1) / RS * RS;
    return (int)(want > kNhRows ? kNhRows : want);
}

static int nh_pick_kt(int vpr) {
    for (int kt = 1; kt <= 4; kt <<= 1)
        if (vpr % kt == 0 && vpr / kt <= kNhBlock) return kt;
    return 0;
}

template <typename T>
static int gn_nhwc_launch(const void* x, void* y, const float* w, const float* b, const float* chan_bias, int stack3,
                          int64_t videos, int T_, int C, int64_t S, int G, float eps, int silu, float* ws, hipStream_t st) {
    constexpr int V = Io<T>::kVec;
    if (C % V) return MVI_EINVAL;
    const int vpr = C / V, kt = nh_pick_kt(vpr);
    if (!kt) return MVI_EINVAL;
    NhGeom q;
    q.R = (int64_t)T_ * S; q.S = S; q.C = C; q.G = G; q.Cg = C / G; q.T = T_;
    q.TPR = vpr / kt; q.RS = kNhBlock / q.TPR;
    q.rpb = nh_rows_per_block(videos, q.R, q.RS);
    q.cpv = (int)((q.R + q.rpb - 1) / q.rpb);
    q.chan_bias = chan_bias; q.stack3 = stack3;
    float* part = ws;
    float* stat = ws + (size_t)videos * q.cpv * G * 3;
    const dim3 grid((unsigned)q.cpv, (unsigned)videos);
    const size_t lds_stats = sizeof(float) * ((size_t)2 * q.RS * C + q.RS), lds_apply = sizeof(float) * 2 * (size_t)C;
    if (lds_stats > 64 * 1024) return MVI_EINVAL;
#define MVI_NH(KT)                                                                                                       \
    hipLaunchKernelGGL((gn_nhwc_stats_kernel<T, KT>), grid, dim3(kNhBlock), lds_stats, st, (const T*)x, part, q);          \
    hipLaunchKernelGGL(gn_nhwc_finalize_kernel, dim3((unsigned)videos, (unsigned)((G + 3) / 4)), dim3(kNhBlock), 0, st, part, stat, q.cpv, G, eps); \
    hipLaunchKernelGGL((gn_nhwc_apply_kernel<T, KT>), grid, dim3(kNhBlock), lds_apply, st, (const T*)x, (T*)y, w, b, stat, q, silu)
    if (kt == 1) { MVI_NH(1); } else if (kt == 2) { MVI_NH(2); } else { MVI_NH(4); }
#undef MVI_NH
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// out[r, c] = h[r, c] + bias[c] (+ x[r, c]) on channels-last rows
template <typename T>
__global__ __launch_bounds__(256) void bias_residual_nhwc_kernel(const T* __restrict__ h, const T* __restrict__ x,
                                                                 const float* __restrict__ bias, T* __restrict__ out,
                                                                 int64_t nvec, int vpr) {
    constexpr int V = Io<T>::kVec;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(v % vpr) * V;
        float a[V];
        Io<T>::load(h + v * V, a);
        if (x) {
            float r[V];
            Io<T>::load(x + v * V, r);
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] += r[k];
        }
        if (bias) {
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] += bias[c0 + k];
        }
        Io<T>::store(out + v * V, a);
    }
}

template <typename T>
static int bias_residual_nhwc_launch(const void* h, const void* x, const float* bias, void* out, int64_t rows, int C, hipStream_t st) {
    constexpr int V = Io<T>::kVec;
    if (C % V) return MVI_EINVAL;
    const int64_t nvec = rows * (C / V);
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL((bias_residual_nhwc_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)h, (const T*)x, bias,
                       (T*)out, nvec, C / V);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi

extern "C" size_t mvi_groupnorm_nhwc_workspace_bytes(int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups) {
    if (videos <= 0 || T <= 0 || C <= 0 || groups <= 0 || spatial <= 0) return 0;
    // sized for the smallest block the launcher may pick (4 rows: one row slot, C >= 1024 vectors wide)
    const int64_t cpv = ((int64_t)T * spatial + 3) / 4;
    return (size_t)(videos * cpv * groups * 3 + videos * groups * 2) * sizeof(float);
}

extern "C" int mvi_groupnorm_silu_nhwc(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                       int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups, float eps,
                                       int32_t fuse_silu, int32_t stack3, int32_t dtype, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    using namespace mvi;
    if (videos < 0 || T <= 0 || C <= 0 || groups <= 0 || spatial < 0 || C % groups != 0)
        return unet_fail(MVI_EINVAL, "groupnorm (channels-last): C must be a positive multiple of groups");
    if (videos == 0 || spatial == 0) return MVI_OK;
    if (!x || !y || !weight || !bias || !workspace) return unet_fail(MVI_EINVAL, "groupnorm (channels-last): NULL pointer");
    if (videos > 65535) return unet_fail(MVI_EINVAL, "groupnorm (channels-last): more than 65535 videos");
    if (stack3 && x == y) return unet_fail(MVI_EINVAL, "groupnorm (channels-last): stack3 output cannot alias the input");
    if (((uintptr_t)x | (uintptr_t)y) % 16 != 0) return unet_fail(MVI_EINVAL, "groupnorm (channels-last): pointers must be 16-byte aligned");
    if (workspace_bytes < mvi_groupnorm_nhwc_workspace_bytes(videos, T, C, spatial, groups))
        return unet_fail(MVI_ENOMEM, "groupnorm (channels-last): workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = gn_nhwc_launch<float>(x, y, weight, bias, chan_bias, stack3 ? 1 : 0, videos, T, C, spatial, groups, eps, fuse_silu, ws, st); break;
        case MVI_DT_BF16: rc = gn_nhwc_launch<__hip_bfloat16>(x, y, weight, bias, chan_bias, stack3 ? 1 : 0, videos, T, C, spatial, groups, eps, fuse_silu, ws, st); break;
        case MVI_DT_F16: rc = gn_nhwc_launch<__half>(x, y, weight, bias, chan_bias, stack3 ? 1 : 0, videos, T, C, spatial, groups, eps, fuse_silu, ws, st); break;
        default: return unet_fail(MVI_EINVAL, "groupnorm (channels-last): unknown dtype");
    }
    if (rc == MVI_EINVAL)
        return unet_fail(MVI_EINVAL, "groupnorm (channels-last): C must be a multiple of the 16-byte vector and at most 1024 vectors wide");
    return rc ? unet_fail(MVI_EHIP, "groupnorm (channels-last): kernel launch failed") : MVI_OK;
}

extern "C" int mvi_groupnorm_nhwc_supported(int32_t C, int32_t groups, int32_t dtype) {
    const int V = dtype == MVI_DT_F32 ? 4 : 8;
    if (C <= 0 || groups <= 0 || C % groups || C % V) return 0;
    const int vpr = C / V, kt = mvi::nh_pick_kt(vpr);
    if (!kt) return 0;
    const int rs = mvi::kNhBlock / (vpr / kt);
    return sizeof(float) * ((size_t)2 * rs * C + rs) <= 64 * 1024 ? 1 : 0;
}

extern "C" int mvi_bias_residual_add_nhwc(const void* h, const void* x, const float* bias, void* out, int64_t rows, int32_t C,
                                          int32_t dtype, void* stream) {
    using namespace mvi;
    if (rows < 0 || C <= 0) return unet_fail(MVI_EINVAL, "bias_residual_add (channels-last): bad shape");
    if (rows == 0) return MVI_OK;
    if (!h || !out) return unet_fail(MVI_EINVAL, "bias_residual_add (channels-last): NULL pointer");
    if (((uintptr_t)h | (uintptr_t)x | (uintptr_t)out) % 16 != 0)
        return unet_fail(MVI_EINVAL, "bias_residual_add (channels-last): pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = bias_residual_nhwc_launch<float>(h, x, bias, out, rows, C, st); break;
        case MVI_DT_BF16: rc = bias_residual_nhwc_launch<__hip_bfloat16>(h, x, bias, out, rows, C, st); break;
        case MVI_DT_F16: rc = bias_residual_nhwc_launch<__half>(h, x, bias, out, rows, C, st); break;
        default: return unet_fail(MVI_EINVAL, "bias_residual_add (channels-last): unknown dtype");
    }
    if (rc == MVI_EINVAL) return unet_fail(MVI_EINVAL, "bias_residual_add (channels-last): C must be a multiple of the 16-byte vector");
    return rc ? unet_fail(MVI_EHIP, "bias_residual_add (channels-last): kernel launch failed") : MVI_OK;
}
