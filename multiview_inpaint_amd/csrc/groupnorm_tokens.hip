// GroupNorm(+SiLU) on TOKEN-MAJOR activations: x [N, S, C] -> y [N, S, C] (C contiguous = NHWC), bf16 / f16 / f32.
// Why it exists: MIOpen runs the 3x3 convolutions of the SVD step with NHWC kernels and, for NCHW tensors, wraps them in
// transposes (9.3 ms per step in `batched_transpose_*`; tools/experiments/conv_layout_probe.py: 817 -> 649 us for one level-0
// convolution, bit-identical output). Inside a ResBlock (openaimodel.py:328-354) the first norm can already write tokens
// (mvi_groupnorm_silu_tokens) and the last add can read them (mvi_tokens_to_planes_add); this is the norm BETWEEN the two
// convolutions, whose input and output are both NHWC. The timestep-embedding bias is fused as in the NCHW kernels (chan_bias).
//
// Three launches, x read twice, y written once (the two-launch NCHW form's traffic):
//   stats : block = VPR x RP threads (VPR = C / vec 16-byte vectors per token, RP rows per pass) over up to 16 chunks of 8 RP tokens;
//           a thread owns the SAME vec channels in every row; shifted sums per channel (see the kernel), assembled into the
//           block's (count, mean, M2) per group;
//   merge : one block per sample: Chan-merges the chunks of every group, then writes per-channel scale / shift
//           (weight * rstd, bias + (chan_bias - mean) * weight * rstd) — 2 C floats per sample;
//   apply : same thread layout as stats, y = act(x * scale[c] + shift[c]).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"
#include "unet_io.h"

namespace mvi {
int unet_fail(int code, const char* msg);

constexpr int kGtPasses = 8;          // token rows per thread and chunk
constexpr int kGtMaxGroups = 64;
constexpr int kGtMergeThreads = 1024;

__host__ __device__ inline int gt_rows_per_pass(int vpr) { int rp = 512 / vpr; return rp < 1 ? 1 : (rp > 8 ? 8 : rp); }

// A block walks `sets` consecutive chunks of 8 rp tokens (a thread owns the SAME vec channels in every row it reads) and keeps, per
// channel, the sums S1 = sum(x - x0) and S2 = sum((x - x0)^2) SHIFTED by the block's first token x0 of that channel — the shifted-data
// form of the variance: S1 and S2 are of the size of the spread, so nothing cancels when the group's moments are assembled from them
// (chan_bias drops out of both; it only moves the mean). One LDS reduction per block instead of six barriers per 8 rp tokens: the
// first form of this pass ran at 2.5 TB/s.
template <typename T>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 8))) void gt_stats_kernel(const T* __restrict__ x, const float* __restrict__ chan_bias, float* __restrict__ part,
                                                        int C, int64_t S, int G, int vpr, int rp, int chunks, int sets) {
    constexpr int V = Io<T>::kVec;
    extern __shared__ float s_mem[];
    float* s_1 = s_mem;                        // [rp][C] S1 per row group, then per channel in row 0
    float* s_2 = s_mem + (size_t)rp * C;       // [rp][C] S2
    float* s_0 = s_mem + (size_t)2 * rp * C;   // [C] x0 + chan_bias
    const int tid = threadIdx.x, v = tid % vpr, r0 = tid / vpr;
    const int64_t n = blockIdx.y;
    const int chunk = blockIdx.x;
    const int rows_set = kGtPasses * rp;
    const int64_t row0 = (int64_t)chunk * rows_set * sets;
    const int64_t row_end = row0 + (int64_t)rows_set * sets < S ? row0 + (int64_t)rows_set * sets : S;
    const int Cg = C / G;
    const T* xb = x + (n * S) * C + (int64_t)v * V;
    float x0[V], s1[V], s2[V];
    Io<T>::load(xb + row0 * C, x0);                              // (row0 < S: the grid covers S)
#pragma unroll
    for (int k = 0; k < V; ++k) s1[k] = s2[k] = 0.f;
    // the rows of a set as loaded (4 registers each: 32 registers of payload, so six to eight waves — and their loads — fit a SIMD;
    // a second set in flight per wave cost more in occupancy than it hid). No load sits in a branch: a row past the block's end
    // reads x0 again (adds 0 to both sums)
    uint4 raw[kGtPasses];
    for (int st = 0; st < sets; ++st) {
        const int64_t base = row0 + (int64_t)st * rows_set;
#pragma unroll
        for (int p = 0; p < kGtPasses; ++p) {
            const int64_t row = base + p * rp + r0;
            raw[p] = *reinterpret_cast<const uint4*>(xb + (row < row_end ? row : row0) * C);
        }
#pragma unroll
        for (int p = 0; p < kGtPasses; ++p) {
            float t[V];
            Io<T>::load(reinterpret_cast<const T*>(&raw[p]), t);
#pragma unroll
            for (int k = 0; k < V; ++k) { const float d = t[k] - x0[k]; s1[k] += d; s2[k] += d * d; }
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        s_1[(size_t)r0 * C + v * V + k] = s1[k];
        s_2[(size_t)r0 * C + v * V + k] = s2[k];
        if (r0 == 0) s_0[v * V + k] = x0[k];
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rp; ++r) { a += s_1[(size_t)r * C + c]; b += s_2[(size_t)r * C + c]; }
        s_1[c] = a;                                              // (row 0 of both: every column is written by the thread that read it)
        s_2[c] = b;
    }
    __syncthreads();
    if (tid < G) {
        const float rows = (float)(row_end - row0);
        const float* cb = chan_bias ? chan_bias + n * C : nullptr;
        float tot = 0.f;
        for (int c = tid * Cg; c < (tid + 1) * Cg; ++c) tot += s_1[c] + rows * (s_0[c] + (cb ? cb[c] : 0.f));
        const float mean = tot / (rows * Cg);
        float m2 = 0.f;
        for (int c = tid * Cg; c < (tid + 1) * Cg; ++c) {
            const float d = mean - (s_0[c] + (cb ? cb[c] : 0.f));          // the group mean seen from this channel's shift
            m2 += s_2[c] - 2.f * d * s_1[c] + rows * d * d;
        }
        float* p = part + ((n * chunks + chunk) * G + tid) * 3;
        p[0] = rows * Cg; p[1] = mean; p[2] = m2 > 0.f ? m2 : 0.f;
    }
}

// Round 6, the token-major residual stream (svd/layers.py TOKEN_STREAM): the elementwise passes that END a block — the ResBlock's skip add
// (openaimodel.py:354), the temporal skip add + AlphaBlender (video_model.py:67-81, util.py:358-372), the transformer's `x + x_in`
// (attention.py:717-722) — on token-major tensors, leaving behind the (count, mean, M2) partials of the GroupNorm that reads the result
// next (the next block's in_layers[0] / SpatialTransformer.norm / the time stack's in_layers[0]) in gt_stats_kernel's layout: that norm then
// runs merge + apply only (mvi_groupnorm_silu_tok2tok_pre) and its statistics pass over the tensor disappears. Thread layout, shifted sums
// and the partials are gt_stats_kernel's; the value whose statistics are taken is the ROUNDED output (what the norm will read).
//   kMode 0: out = a + b + bias[c]                       (b and / or bias may be NULL)
//   kMode 1: out = base + (1 - alpha[n]) * (a + bias[c]) (alpha per sample: tokens_blend_to_planes_kernel's arithmetic)
//   kMode 2: out = concat_channels(a [.., C1], b [.., C - C1] + base)   (the decoder's th.cat([h, hs.pop() + control.pop()], dim=1),
//            csvd.py:84-91; base may be NULL) — the statistics are those of GroupNorm(G) over the C concatenated channels
// G = 0: no statistics (the plain fused pass).
template <typename T, int kMode>
__global__ __launch_bounds__(1024) void gt_fused_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ base,
                                                        const float* __restrict__ bias, const float* __restrict__ alpha, T* __restrict__ out,
                                                        float* __restrict__ part, int C, int64_t S, int G, int vpr, int rp, int chunks, int sets,
                                                        int C1, int parts) {
    constexpr int V = Io<T>::kVec;
    extern __shared__ float s_mem[];
    float* s_1 = s_mem;                        // [rp][C], as in gt_stats_kernel
    float* s_2 = s_mem + (size_t)rp * C;
    float* s_0 = s_mem + (size_t)2 * rp * C;   // [C] the block's shift: its first row's OUTPUT
    const int tid = threadIdx.x, v = tid % vpr, r0 = tid / vpr;
    const int64_t n = blockIdx.y;
    const int chunk = blockIdx.x;
    const int rows_set = kGtPasses * rp;
    const int64_t row0 = (int64_t)chunk * rows_set * sets;
    const int64_t row_end = row0 + (int64_t)rows_set * sets < S ? row0 + (int64_t)rows_set * sets : S;
    const int64_t off = (n * S) * C + (int64_t)v * V;
    // this thread's operands: first (always there) and second (may be absent), each with its own row pitch
    const T* first = a + off;
    const T* second = (kMode == 1 ? base : b) ? (kMode == 1 ? base : b) + off : nullptr;
    int pitch = C;
    if (kMode == 2) {
        const bool left = v * V < C1;
        pitch = left ? C1 : C - C1;
        const int64_t o2 = (n * S) * pitch + (int64_t)v * V - (left ? 0 : C1);
        first = (left ? a : b) + o2;
        second = !left && base ? base + o2 : nullptr;
    }
    float bv[V];
#pragma unroll
    for (int k = 0; k < V; ++k) bv[k] = bias ? bias[v * V + k] : 0.f;
    const float one_minus = kMode == 1 ? 1.0f - alpha[n] : 0.f;
    auto value = [&](const uint4& ra, const uint4& rb, float (&o)[V]) __attribute__((always_inline)) {
        float ta[V], tb[V];
        Io<T>::load(reinterpret_cast<const T*>(&ra), ta);
        Io<T>::load(reinterpret_cast<const T*>(&rb), tb);
#pragma unroll
        for (int k = 0; k < V; ++k)
            o[k] = round_to<T>(kMode == 1 ? tb[k] + one_minus * (ta[k] + bv[k]) : (second ? ta[k] + bv[k] + tb[k] : ta[k] + bv[k]));
    };
    // the shift: the output of the block's first row, computed by every thread for its own channels (rows past the block's end are
    // replaced by that row below: they add 0 to both sums and are not stored)
    float x0[V], s1[V], s2[V];
    {
        const uint4 ra = *reinterpret_cast<const uint4*>(first + row0 * pitch);
        const uint4 rb = second ? *reinterpret_cast<const uint4*>(second + row0 * pitch) : make_uint4(0, 0, 0, 0);
        value(ra, rb, x0);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) s1[k] = s2[k] = 0.f;
    constexpr int kHalf = kGtPasses / 2;                              // four rows in flight per operand
    for (int st = 0; st < 2 * sets; ++st) {
        const int64_t base_row = row0 + (int64_t)st * kHalf * rp;
        uint4 ra[kHalf], rb[kHalf];
#pragma unroll
        for (int p = 0; p < kHalf; ++p) {
            const int64_t row = base_row + p * rp + r0;
            const int64_t rc = row < row_end ? row : row0;
            ra[p] = *reinterpret_cast<const uint4*>(first + rc * pitch);
            rb[p] = second ? *reinterpret_cast<const uint4*>(second + rc * pitch) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < kHalf; ++p) {
            const int64_t row = base_row + p * rp + r0;
            float o[V];
            value(ra[p], rb[p], o);
            if (row < row_end) Io<T>::store(out + off + row * C, o);
#pragma unroll
            for (int k = 0; k < V; ++k) { const float d = o[k] - x0[k]; s1[k] += d; s2[k] += d * d; }
        }
    }
    if (G == 0) return;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        s_1[(size_t)r0 * C + v * V + k] = s1[k];
        s_2[(size_t)r0 * C + v * V + k] = s2[k];
        if (r0 == 0) s_0[v * V + k] = x0[k];
    }
    __syncthreads();
    // The block's (count, mean, M2) per group, in two short steps instead of one thread per group walking its C / G channels twice (at
    // 1280 channels that walk was a third of the kernel): thread (g, part) merges the C / (G parts) channels of its part — a channel's
    // `rows` values have mean x0 + S1 / rows and M2 = S2 - S1^2 / rows (shifted sums: nothing cancels) — by Chan's update for sets of
    // equal size, then one thread per group merges the parts in a fixed order.
    float* s_q = s_0 + C;                      // [G * parts][2]
    const int Cg = C / G, Cp = Cg / parts;
    const float rows = (float)(row_end - row0), inv_rows = 1.0f / rows;
    if (tid < G * parts) {
        const int c0 = tid / parts * Cg + tid % parts * Cp;
        float mean = 0.f, m2 = 0.f;
        for (int k = 0; k < Cp; ++k) {
            float a1 = 0.f, a2 = 0.f;
            for (int r = 0; r < rp; ++r) { a1 += s_1[(size_t)r * C + c0 + k]; a2 += s_2[(size_t)r * C + c0 + k]; }
            const float mc = s_0[c0 + k] + a1 * inv_rows, qc = a2 - a1 * a1 * inv_rows;
            const float d = mc - mean, inv = 1.0f / (float)(k + 1);
            mean += d * inv;
            m2 += qc + d * d * (rows * (float)k * inv);
        }
        s_q[2 * tid] = mean; s_q[2 * tid + 1] = m2;
    }
    __syncthreads();
    if (tid < G) {
        const float n_part = rows * (float)Cp;
        float mean = 0.f, m2 = 0.f;
        for (int k = 0; k < parts; ++k) {
            const float mb = s_q[2 * (tid * parts + k)], qb = s_q[2 * (tid * parts + k) + 1];
            const float d = mb - mean, inv = 1.0f / (float)(k + 1);
            mean += d * inv;
            m2 += qb + d * d * (n_part * (float)k * inv);
        }
        float* p = part + ((n * chunks + chunk) * G + tid) * 3;
        p[0] = rows * Cg; p[1] = mean; p[2] = m2 > 0.f ? m2 : 0.f;
    }
}

// one block per sample: Chan merge per group, then per-channel scale / shift
// frames > 1: the temporal GroupNorm of VideoResBlock.time_stack (statistics over the `frames` consecutive samples of a video,
// video_model.py:71-75): a block merges the chunks of all its frames — they are consecutive in `part` — and writes every frame's
// scale / shift (the frames differ in chan_bias only).
__global__ __launch_bounds__(kGtMergeThreads) void gt_merge_kernel(const float* __restrict__ part, const float* __restrict__ weight, const float* __restrict__ bias,
                                                       const float* __restrict__ chan_bias, float* __restrict__ scale_shift, int C, int G,
                                                       int chunks_per_frame, float eps, int frames) {
    __shared__ float s_mean[kGtMaxGroups], s_rstd[kGtMaxGroups];
    __shared__ float s_p[kGtMergeThreads][3];
    // one block per FRAME: it merges the chunks of its whole video (redundantly with its frames - 1 siblings: a few hundred partials
    // from L2) and writes its own frame's scale / shift — one block per video left 2 blocks walking frames x C channels (25-37 us)
    const int64_t frame = blockIdx.x;
    const int64_t n = frame / frames * frames;              // first frame of the video
    const int chunks = chunks_per_frame * frames;
    const int tid = threadIdx.x;
    // 1024 / G threads per group take the chunks round-robin, four loads ahead of the merge chain (a walk with one dependent load per
    // step costs a memory latency per chunk: 20-37 us per call with hundreds of chunks per group), then one thread per group merges
    // their partials in a fixed order
    const int per = kGtMergeThreads / G, g = tid % G, sub = tid / G;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    if (sub < per) {
        for (int c = sub; c < chunks; c += 4 * per) {
            float nb[4], mb[4], qb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = c + u * per;
                const float* p = part + ((n * chunks_per_frame + (cc < chunks ? cc : c)) * G + g) * 3;
                nb[u] = cc < chunks ? p[0] : 0.f; mb[u] = p[1]; qb[u] = p[2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (nb[u] > 0.f) {
                    const float nt = cnt + nb[u], d = mb[u] - mean;
                    mean += d * (nb[u] / nt);
                    m2 += qb[u] + d * d * (cnt * nb[u] / nt);
                    cnt = nt;
                }
            }
        }
    }
    s_p[tid][0] = cnt; s_p[tid][1] = mean; s_p[tid][2] = m2;
    __syncthreads();
    if (tid < G) {
        cnt = 0.f; mean = 0.f; m2 = 0.f;
        for (int k = 0; k < per; ++k) {
            const float nb = s_p[k * G + tid][0], mb = s_p[k * G + tid][1], qb = s_p[k * G + tid][2];
            if (nb > 0.f) {
                const float nt = cnt + nb, d = mb - mean;
                mean += d * (nb / nt);
                m2 += qb + d * d * (cnt * nb / nt);
                cnt = nt;
            }
        }
        s_mean[tid] = mean;
        s_rstd[tid] = rsqrtf(m2 / cnt + eps);
    }
    __syncthreads();
    const int Cg = C / G;
    for (int c = tid; c < C; c += kGtMergeThreads) {
        const int g = c / Cg;
        const float w = weight[c] * s_rstd[g];
        const float add = chan_bias ? chan_bias[frame * C + c] : 0.f;
        scale_shift[(frame * C + c) * 2 + 0] = w;
        scale_shift[(frame * C + c) * 2 + 1] = bias[c] + (add - s_mean[g]) * w;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void gt_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scale_shift, int C,
                                                        int64_t S, int vpr, int rp, int silu) {
    constexpr int V = Io<T>::kVec;
    const int tid = threadIdx.x, v = tid % vpr, r0 = tid / vpr;
    const int64_t n = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * (kGtPasses * rp);
    const float* ss = scale_shift + (n * C + v * V) * 2;
    float sc[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sc[k] = ss[2 * k]; sh[k] = ss[2 * k + 1]; }
    const T* xb = x + (n * S) * C + (int64_t)v * V;
    T* yb = y + (n * S) * C + (int64_t)v * V;
    uint4 raw[kGtPasses];                       // as loaded (4 registers per row, decoded one row at a time): see gt_stats_kernel
#pragma unroll
    for (int p = 0; p < kGtPasses; ++p) {       // (no load in a branch: rows past the end read the last row and are not stored)
        const int64_t row = row0 + p * rp + r0;
        raw[p] = *reinterpret_cast<const uint4*>(xb + (row < S ? row : S - 1) * C);
    }
#pragma unroll
    for (int p = 0; p < kGtPasses; ++p) {
        const int64_t row = row0 + p * rp + r0;
        if (row < S) {
            float t[V];
            Io<T>::load(reinterpret_cast<const T*>(&raw[p]), t);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float u = t[k] * sc[k] + sh[k];
                t[k] = silu ? u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f)) : u;
            }
            Io<T>::store(yb + row * C, t);
        }
    }
}

// Round 6, the split-operand convolutions of the first-stage decoder (csrc/linear_n320.hip, mvi_conv3x3_split3_f32): the apply pass of an
// fp32 tensor writes every value as TWO bf16 — y2 [N, S, 2 C] = (hi | lo), hi = round(v), lo = round(v - hi): 16 mantissa bits, the bytes
// of the fp32 tensor — instead of one rounded value. scale_shift NULL: no normalisation (the plain split in front of Upsample.conv).
// kMode 0: (hi | lo) bf16, rows of 2 C; 1: hi only, bf16, rows of C; 2: one f16 value, rows of C (the operands of the terms = 1 launches).
template <int kMode>
__global__ __launch_bounds__(1024) void gt_apply_split_kernel(const float* __restrict__ x, uint16_t* __restrict__ y2, const float* __restrict__ scale_shift,
                                                              int C, int64_t S, int vpr, int rp, int silu) {
    constexpr int V = 4;
    const int tid = threadIdx.x, v = tid % vpr, r0 = tid / vpr;
    const int64_t n = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * (kGtPasses * rp);
    float sc[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        sc[k] = scale_shift ? scale_shift[(n * C + v * V + k) * 2] : 1.f;
        sh[k] = scale_shift ? scale_shift[(n * C + v * V + k) * 2 + 1] : 0.f;
    }
    const float* xb = x + (n * S) * C + (int64_t)v * V;
    constexpr int kRow = kMode == 0 ? 2 : 1;                       // output row = kRow C elements
    uint16_t* yb = y2 + (n * S) * kRow * C + (int64_t)v * V;
    float4 raw[kGtPasses];
#pragma unroll
    for (int p = 0; p < kGtPasses; ++p) {
        const int64_t row = row0 + p * rp + r0;
        raw[p] = *reinterpret_cast<const float4*>(xb + (row < S ? row : S - 1) * C);
    }
#pragma unroll
    for (int p = 0; p < kGtPasses; ++p) {
        const int64_t row = row0 + p * rp + r0;
        if (row < S) {
            const float t[V] = {raw[p].x, raw[p].y, raw[p].z, raw[p].w};
            uint16_t hi[V], lo[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float u = t[k] * sc[k] + sh[k];
                if (silu) u = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
                if (kMode == 2) {
                    const __half h = __float2half(u);
                    hi[k] = *reinterpret_cast<const uint16_t*>(&h);
                    lo[k] = 0;
                } else {
                    const __hip_bfloat16 h = __float2bfloat16(u);
                    const __hip_bfloat16 l = __float2bfloat16(u - __bfloat162float(h));
                    hi[k] = *reinterpret_cast<const uint16_t*>(&h);
                    lo[k] = *reinterpret_cast<const uint16_t*>(&l);
                }
            }
            const uint2 ph = {(uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16)};
            *reinterpret_cast<uint2*>(yb + row * kRow * C) = ph;
            if (kMode == 0) {
                const uint2 pl = {(uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16)};
                *reinterpret_cast<uint2*>(yb + row * 2 * C + C) = pl;
            }
        }
    }
}

static int gt_launch_split(const float* x, void* y2, const float* w, const float* b, const float* cb, int64_t N, int C, int64_t S, int G, float eps,
                           int silu, float* ws, hipStream_t st, int frames, int mode) {
    constexpr int V = 4;
    const int vpr = C / V, rp = gt_rows_per_pass(vpr);
    const int chunks = (int)((S + kGtPasses * rp - 1) / (kGtPasses * rp));
    const dim3 grid((unsigned)chunks, (unsigned)N), block((unsigned)(vpr * rp));
    float* ss = nullptr;
    if (G > 0) {
        int sets = (int)((int64_t)chunks * N / 768);
        sets = sets < 1 ? 1 : (sets > 16 ? 16 : sets);
        const int schunks = (chunks + sets - 1) / sets;
        float* part = ws;
        ss = ws + (size_t)N * chunks * G * 3;
        const size_t lds = ((size_t)2 * rp * C + C) * sizeof(float);
        hipLaunchKernelGGL((gt_stats_kernel<float>), dim3((unsigned)schunks, (unsigned)N), block, lds, st, x, cb, part, C, S, G, vpr, rp, schunks, sets);
        hipLaunchKernelGGL(gt_merge_kernel, dim3((unsigned)N), dim3(kGtMergeThreads), 0, st, part, w, b, cb, ss, C, G, schunks, eps, frames);
    }
    if (mode == 0) hipLaunchKernelGGL(gt_apply_split_kernel<0>, grid, block, 0, st, x, (uint16_t*)y2, ss, C, S, vpr, rp, silu);
    else if (mode == 1) hipLaunchKernelGGL(gt_apply_split_kernel<1>, grid, block, 0, st, x, (uint16_t*)y2, ss, C, S, vpr, rp, silu);
    else hipLaunchKernelGGL(gt_apply_split_kernel<2>, grid, block, 0, st, x, (uint16_t*)y2, ss, C, S, vpr, rp, silu);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

template <typename T>
static int gt_launch(const void* x, void* y, const float* w, const float* b, const float* cb, int64_t N, int C, int64_t S, int G, float eps,
                     int silu, float* ws, hipStream_t st, int frames = 1) {
    constexpr int V = Io<T>::kVec;
    const int vpr = C / V, rp = gt_rows_per_pass(vpr);
    const int chunks = (int)((S + kGtPasses * rp - 1) / (kGtPasses * rp));
    // the statistics pass walks `sets` chunks per block: as many as leave ~3 blocks per CU (at most 16)
    int sets = (int)((int64_t)chunks * N / 768);
    sets = sets < 1 ? 1 : (sets > 16 ? 16 : sets);
    const int schunks = (chunks + sets - 1) / sets;
    float* part = ws;
    float* ss = ws + (size_t)N * chunks * G * 3;              // (sized for sets = 1)
    const dim3 grid((unsigned)chunks, (unsigned)N), block((unsigned)(vpr * rp));
    const size_t lds = ((size_t)2 * rp * C + C) * sizeof(float);
    hipLaunchKernelGGL((gt_stats_kernel<T>), dim3((unsigned)schunks, (unsigned)N), block, lds, st, (const T*)x, cb, part, C, S, G, vpr, rp, schunks, sets);
    hipLaunchKernelGGL(gt_merge_kernel, dim3((unsigned)N), dim3(kGtMergeThreads), 0, st, part, w, b, cb, ss, C, G, schunks, eps, frames);
    hipLaunchKernelGGL((gt_apply_kernel<T>), grid, block, 0, st, (const T*)x, (T*)y, ss, C, S, vpr, rp, silu);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// merge + apply with the statistics ALREADY in `part` [N * chunks_per_sample * G][3] — left there by the producer of x (the
// implicit-GEMM convolution's epilogue, csrc/linear_n320.hip kStats): x is read once, y written once
template <typename T>
static int gt_launch_pre(const void* x, void* y, const float* w, const float* b, const float* cb, int64_t N, int C, int64_t S, int G, float eps,
                         int silu, const float* part, int chunks_per_sample, float* ws, hipStream_t st, int frames) {
    constexpr int V = Io<T>::kVec;
    const int vpr = C / V, rp = gt_rows_per_pass(vpr);
    const int chunks = (int)((S + kGtPasses * rp - 1) / (kGtPasses * rp));
    float* ss = ws;                                               // [N, C, 2]
    const dim3 grid((unsigned)chunks, (unsigned)N), block((unsigned)(vpr * rp));
    hipLaunchKernelGGL(gt_merge_kernel, dim3((unsigned)N), dim3(kGtMergeThreads), 0, st, part, w, b, cb, ss, C, G, chunks_per_sample, eps, frames);
    hipLaunchKernelGGL((gt_apply_kernel<T>), grid, block, 0, st, (const T*)x, (T*)y, ss, C, S, vpr, rp, silu);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi

static int gt_geometry_ok(int64_t N, int32_t C, int64_t S, int32_t G, int32_t dtype) {
    const int V = dtype == MVI_DT_F32 ? 4 : 8;
    if (N <= 0 || C <= 0 || S <= 0 || G <= 0 || G > mvi::kGtMaxGroups || C % G || C % V) return 0;
    const int vpr = C / V;
    if (vpr > 1024 || N > 65535) return 0;
    const int rp = mvi::gt_rows_per_pass(vpr);
    return ((size_t)2 * rp * C + C) * sizeof(float) <= 64 * 1024;
}

extern "C" size_t mvi_groupnorm_tok2tok_workspace_bytes(int64_t N, int32_t C, int64_t spatial, int32_t groups, int32_t dtype) {
    if (!gt_geometry_ok(N, C, spatial, groups, dtype)) return 0;
    const int V = dtype == MVI_DT_F32 ? 4 : 8;
    const int rp = mvi::gt_rows_per_pass(C / V);
    const int64_t chunks = (spatial + mvi::kGtPasses * rp - 1) / (mvi::kGtPasses * rp);
    return (size_t)(N * chunks * groups * 3 + N * C * 2) * sizeof(float);
}

static int tok2tok_impl(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias, int64_t N, int32_t C,
                        int64_t spatial, int32_t groups, float eps, int32_t fuse_silu, int32_t frames, int32_t dtype, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (N == 0 || spatial == 0) return MVI_OK;
    if (!gt_geometry_ok(N, C, spatial, groups, dtype))
        return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok: C must be a multiple of groups (<= 64) and of the 16-byte vector width");
    if (frames < 1 || N % frames) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok: N must be a whole number of videos of `frames` samples");
    if (!x || !y || !weight || !bias || !workspace) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok: NULL pointer");
    if (((uintptr_t)x | (uintptr_t)y) % 16) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok: x / y must be 16-byte aligned");
    if (workspace_bytes < mvi_groupnorm_tok2tok_workspace_bytes(N, C, spatial, groups, dtype))
        return mvi::unet_fail(MVI_ENOMEM, "groupnorm_tok2tok: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::gt_launch<float>(x, y, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, (float*)workspace, st, frames); break;
        case MVI_DT_BF16: rc = mvi::gt_launch<__hip_bfloat16>(x, y, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, (float*)workspace, st, frames); break;
        case MVI_DT_F16: rc = mvi::gt_launch<__half>(x, y, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, (float*)workspace, st, frames); break;
        default: return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok: unknown dtype");
    }
    return rc ? mvi::unet_fail(MVI_EHIP, "groupnorm_tok2tok: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_groupnorm_silu_tok2tok_pre(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                              int64_t N, int32_t frames, int32_t C, int64_t spatial, int32_t groups, float eps,
                                              int32_t fuse_silu, int32_t dtype, const float* part, int32_t chunks_per_sample,
                                              void* workspace, size_t workspace_bytes, void* stream) {
    if (N == 0 || spatial == 0) return MVI_OK;
    if (!gt_geometry_ok(N, C, spatial, groups, dtype) || dtype == MVI_DT_F32)
        return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_pre: bf16 / f16, C a multiple of groups (<= 64) and of the 16-byte vector width");
    if (frames < 1 || N % frames || chunks_per_sample < 1) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_pre: bad frames / chunks");
    if (!x || !y || !weight || !bias || !part || !workspace) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_pre: NULL pointer");
    if (((uintptr_t)x | (uintptr_t)y) % 16) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_pre: x / y must be 16-byte aligned");
    if (workspace_bytes < (size_t)N * C * 2 * sizeof(float)) return mvi::unet_fail(MVI_ENOMEM, "groupnorm_tok2tok_pre: workspace too small (N C 2 floats)");
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == MVI_DT_BF16
                       ? mvi::gt_launch_pre<__hip_bfloat16>(x, y, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, part, chunks_per_sample, (float*)workspace, st, frames)
                       : mvi::gt_launch_pre<__half>(x, y, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, part, chunks_per_sample, (float*)workspace, st, frames);
    return rc ? mvi::unet_fail(MVI_EHIP, "groupnorm_tok2tok_pre: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_groupnorm_silu_tok2tok(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                          int64_t N, int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu,
                                          int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return tok2tok_impl(x, y, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, 1, dtype, workspace, workspace_bytes, stream);
}

extern "C" int mvi_groupnorm_silu_tok2tok_frames(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                                 int64_t N, int32_t frames, int32_t C, int64_t spatial, int32_t groups, float eps,
                                                 int32_t fuse_silu, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return tok2tok_impl(x, y, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, frames, dtype, workspace, workspace_bytes, stream);
}

// GroupNorm(+SiLU) of an fp32 token-major tensor x [N, S, C] written as split bf16 y2 [N, S, 2 C] = (hi | lo) (see gt_apply_split_kernel);
// statistics over the `frames` consecutive samples of a video when frames > 1 (the temporal norms). groups = 0: no normalisation, the
// plain split (weight / bias / chan_bias / workspace unused). Workspace as mvi_groupnorm_tok2tok_workspace_bytes(..., MVI_DT_F32).
// out_mode 1 / 2: ONE rounded value per element instead, bf16 / f16, y2 [N, S, C] (the operands of the terms = 1 convolutions).
extern "C" int mvi_groupnorm_silu_tok2tok_split(const float* x, void* y2, const float* weight, const float* bias, const float* chan_bias,
                                                int64_t N, int32_t frames, int32_t C, int64_t spatial, int32_t groups, float eps,
                                                int32_t fuse_silu, int32_t out_mode, void* workspace, size_t workspace_bytes, void* stream) {
    if (N == 0 || spatial == 0) return MVI_OK;
    if (out_mode < 0 || out_mode > 2) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_split: out_mode 0 (hi | lo bf16), 1 (bf16) or 2 (f16)");
    if (!gt_geometry_ok(N, C, spatial, groups > 0 ? groups : 1, MVI_DT_F32))
        return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_split: C must be a multiple of groups (<= 64) and of 4");
    if (frames < 1 || N % frames) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_split: N must be a whole number of videos of `frames` samples");
    if (!x || !y2 || (groups > 0 && (!weight || !bias || !workspace))) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_split: NULL pointer");
    if (((uintptr_t)x | (uintptr_t)y2) % 16 || C % 4) return mvi::unet_fail(MVI_EINVAL, "groupnorm_tok2tok_split: x / y2 must be 16-byte aligned");
    if (groups > 0 && workspace_bytes < mvi_groupnorm_tok2tok_workspace_bytes(N, C, spatial, groups, MVI_DT_F32))
        return mvi::unet_fail(MVI_ENOMEM, "groupnorm_tok2tok_split: workspace too small");
    const int rc = mvi::gt_launch_split(x, y2, weight, bias, chan_bias, N, C, spatial, groups, eps, fuse_silu, (float*)workspace, (hipStream_t)stream, frames, out_mode);
    return rc ? mvi::unet_fail(MVI_EHIP, "groupnorm_tok2tok_split: kernel launch failed") : MVI_OK;
}

// ---- round 6: elementwise block tails on token-major tensors with the NEXT GroupNorm's statistics (gt_fused_kernel) --------------------
namespace mvi {
template <typename T>
static int gt_fused_launch(int mode, const void* a, const void* b, const void* base, const float* bias, const float* alpha, void* out, float* part,
                           int64_t N, int C, int64_t S, int G, int* chunks_out, hipStream_t st, int C1) {
    constexpr int V = Io<T>::kVec;
    const int vpr = C / V, rp = gt_rows_per_pass(vpr);
    const int chunks = (int)((S + kGtPasses * rp - 1) / (kGtPasses * rp));
    // blocks walk `sets` chunks each: ~6 blocks per CU (level 0 of the 576x1024 step, [28, 9216, 320]: 113 us at 3 per CU, 101 us at 6,
    // 96 us at 12 — but every chunk is a partial the norm's merge walks, 14 frames' worth for the temporal norm)
    int sets = (int)((int64_t)chunks * N / 1536);
    sets = sets < 1 ? 1 : (sets > 16 ? 16 : sets);
    const int schunks = (chunks + sets - 1) / sets;
    if (chunks_out) *chunks_out = schunks;
    // parts per group of the block's closing merge: the largest divisor of C / G that is <= 16 and leaves G parts <= the block's threads
    int parts = 1;
    if (G > 0)
        for (int q = 2; q <= 16 && G * q <= vpr * rp; ++q)
            if ((C / G) % q == 0) parts = q;
    const size_t lds = G > 0 ? ((size_t)2 * rp * C + C + 2 * G * parts) * sizeof(float) : 0;
    const dim3 grid((unsigned)schunks, (unsigned)N), block((unsigned)(vpr * rp));
    if (mode == 0)
        hipLaunchKernelGGL((gt_fused_kernel<T, 0>), grid, block, lds, st, (const T*)a, (const T*)b, (const T*)nullptr, bias, alpha, (T*)out, part, C, S, G, vpr, rp, schunks, sets, 0, parts);
    else if (mode == 1)
        hipLaunchKernelGGL((gt_fused_kernel<T, 1>), grid, block, lds, st, (const T*)a, (const T*)nullptr, (const T*)base, bias, alpha, (T*)out, part, C, S, G, vpr, rp, schunks, sets, 0, parts);
    else
        hipLaunchKernelGGL((gt_fused_kernel<T, 2>), grid, block, lds, st, (const T*)a, (const T*)b, (const T*)base, (const float*)nullptr, alpha, (T*)out, part, C, S, G, vpr, rp, schunks, sets, C1, parts);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
}  // namespace mvi

// Partials buffer of the fused passes below: [N * chunks * groups][3] floats with chunks <= the statistics pass's chunk count.
extern "C" size_t mvi_rows_gnstats_bytes(int64_t N, int32_t C, int64_t spatial, int32_t groups, int32_t dtype) {
    if (groups <= 0 || !gt_geometry_ok(N, C, spatial, groups, dtype)) return 0;
    const int V = dtype == MVI_DT_F32 ? 4 : 8;
    const int rp = mvi::gt_rows_per_pass(C / V);
    if (((size_t)2 * rp * C + C + 2 * groups * 16) * sizeof(float) > 64 * 1024) return 0;      // (+ the parts of the closing merge)
    const int64_t chunks = (spatial + mvi::kGtPasses * rp - 1) / (mvi::kGtPasses * rp);
    return (size_t)(N * chunks * groups * 3) * sizeof(float);
}

// mode 0: out = a + b + bias[c] (b, bias optional); mode 1: out = base + (1 - alpha[n]) * (a + bias[c]) — token-major [N, spatial, C] tensors
// of one dtype (bf16 / f16 / fp32), bias [C] / alpha [N] fp32; mode 2: out [N, spatial, C] = channels (a [.., C_first] | b [.., C - C_first]
// + base), base optional, no bias (C_first is ignored by the other modes). groups > 0: also the (count, mean, M2) partials of GroupNorm(groups) of `out`
// into part (mvi_rows_gnstats_bytes), *chunks_per_sample = their chunk count — what mvi_groupnorm_silu_tok2tok_pre takes; groups = 0: none.
extern "C" int mvi_rows_fused_gnstats(int32_t mode, const void* a, const void* b, const void* base, const float* bias, const float* alpha,
                                      void* out, int64_t N, int32_t C, int32_t C_first, int64_t spatial, int32_t groups, int32_t dtype, float* part,
                                      size_t part_bytes, int32_t* chunks_per_sample, void* stream) {
    if (N == 0 || spatial == 0) return MVI_OK;
    if (mode < 0 || mode > 2) return mvi::unet_fail(MVI_EINVAL, "rows_fused_gnstats: mode 0 (add), 1 (blend) or 2 (concat)");
    if (mode == 2 && (!b || bias || C_first <= 0 || C_first >= C || C_first % (dtype == MVI_DT_F32 ? 4 : 8) || (C - C_first) % (dtype == MVI_DT_F32 ? 4 : 8)))
        return mvi::unet_fail(MVI_EINVAL, "rows_fused_gnstats: concat takes a [.., C_first] and b [.., C - C_first], both multiples of the 16-byte vector, and no bias");
    if (!gt_geometry_ok(N, C, spatial, groups > 0 ? groups : 1, dtype))
        return mvi::unet_fail(MVI_EINVAL, "rows_fused_gnstats: C must be a multiple of groups (<= 64) and of the 16-byte vector width");
    if (!a || !out || (mode == 1 && (!base || !alpha))) return mvi::unet_fail(MVI_EINVAL, "rows_fused_gnstats: NULL pointer");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)base | (uintptr_t)out) % 16) return mvi::unet_fail(MVI_EINVAL, "rows_fused_gnstats: tensors must be 16-byte aligned");
    if (groups > 0 && (!part || part_bytes < mvi_rows_gnstats_bytes(N, C, spatial, groups, dtype)))
        return mvi::unet_fail(MVI_EINVAL, "rows_fused_gnstats: statistics buffer missing or too small (mvi_rows_gnstats_bytes)");
    hipStream_t st = (hipStream_t)stream;
    int rc, ch = 0;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::gt_fused_launch<float>(mode, a, b, base, bias, alpha, out, part, N, C, spatial, groups, &ch, st, C_first); break;
        case MVI_DT_BF16: rc = mvi::gt_fused_launch<__hip_bfloat16>(mode, a, b, base, bias, alpha, out, part, N, C, spatial, groups, &ch, st, C_first); break;
        case MVI_DT_F16: rc = mvi::gt_fused_launch<__half>(mode, a, b, base, bias, alpha, out, part, N, C, spatial, groups, &ch, st, C_first); break;
        default: return mvi::unet_fail(MVI_EINVAL, "rows_fused_gnstats: unknown dtype");
    }
    if (chunks_per_sample) *chunks_per_sample = ch;
    return rc ? mvi::unet_fail(MVI_EHIP, "rows_fused_gnstats: kernel launch failed") : MVI_OK;
}
