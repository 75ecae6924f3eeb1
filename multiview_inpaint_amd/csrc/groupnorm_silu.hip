// Fused GroupNorm(+SiLU) for gfx950 — HBM-bound: 2 reads + 1 write of the tensor.
// Replaces nn.GroupNorm -> nn.SiLU pairs of the reference UNet
// (svd_inpaint1/sgm/modules/diffusionmodules/openaimodel.py:257-261,292-305; GroupNorm32 fp32
// upcast util.py:274-276; attention.py:125-128 for the eps=1e-6, no-SiLU stem norm).
//
// x [N, C, S] contiguous: one group = Cg*S contiguous elements. Two launches:
//   stats : grid (chunks, N*G); each 256-thread block holds its chunk in registers, computes the
//           chunk mean and the centred second moment M2 exactly (two passes over registers) and
//           writes (count, mean, M2);
//   apply : same grid; every block merges its group's partials with Chan's formula (fp32), then
//           normalises its chunk with per-channel affine (+SiLU) and writes y in x's dtype.
// Loads/stores are 16 B per lane. The chunk partition is identical in both launches.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"
#include "unet_io.h"

namespace mvi {

constexpr int kGnBlock = 256;
constexpr int kGnVecPerThread = 8;                        // 8 x 16 B per thread

// e / S for 0 <= e < 2^24 (an element index inside one group slice) without the 64-bit integer division the plain
// expression costs (~100 instructions per vector): reciprocal estimate, then one step of correction either way
__device__ __forceinline__ int div_small(uint32_t e, uint32_t S, float inv_S) {
    uint32_t qd = (uint32_t)((float)e * inv_S);
    if (qd * S > e) --qd;
    if ((qd + 1) * S <= e) ++qd;
    return (int)qd;
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// chunk = kGnBlock * kGnVecPerThread * kVec elements of one group (vector path needs E % kVec == 0
// and a 16-B aligned base, checked on the host; otherwise VEC = false walks scalars)
// A group is `slices` pieces of E contiguous elements each, piece s of group (n, g) starting at
// ((n * slices + s) * C + g * Cg) * S. slices == 1 is the ordinary [N, C, S] GroupNorm; slices == T is
// the temporal GroupNorm over "b c t h w" evaluated directly on the "(b t) c h w" tensor, no permute.
// grid.x = slices * cps (chunks per slice), grid.y = N * G.
struct GnGeom {
    int64_t E;        // elements per slice = Cg * S
    int64_t S;        // spatial positions per channel in a slice
    int64_t slice_stride;   // C * S
    int cps, slices, Cg, G;
    const float* chan_bias;   // optional [rows, C] added to x before the statistics (row = n * slices + slice):
                              // the timestep-embedding bias of ResBlock (openaimodel.py:341-352) fused into the norm
    int stack3;               // temporal form only: write y three times into [(videos T), 3C, S] — at channel
                              // offset C of its own frame, offset 0 of the next frame and offset 2C of the previous
                              // one (zeros at the sequence ends): the input of the (3,1,1) convolution as a 1x1
};
__device__ __forceinline__ int64_t gn_slice_base(const GnGeom& q, int64_t g, int slice) {
    const int64_t n = g / q.G, gi = g % q.G;
    return (n * q.slices + slice) * q.slice_stride + gi * q.E;
}

// The VEC form keeps the chunk AS LOADED (16 bytes = 4 registers per vector instead of 8 floats) and decodes it once per pass:
// 32 registers of payload instead of 64, 8 waves per SIMD instead of 5 — the pass is a read-only stream whose rate follows the
// number of loads a CU has in flight (same change in gt_stats_kernel: 140 -> 119 us at (28, 320, 72, 128)).
template <typename T, bool VEC>
__global__ __launch_bounds__(kGnBlock) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, GnGeom q) {
    __shared__ float s_red[4];
    constexpr int KV = Io<T>::kVec;
    constexpr int CH = kGnBlock * kGnVecPerThread * KV;
    const int64_t g = blockIdx.y;
    const int slice = blockIdx.x / q.cps, chunk = blockIdx.x % q.cps;
    const int64_t E = q.E;
    const int chunks = q.cps * q.slices;
    const int64_t e0 = (int64_t)chunk * CH;
    const T* base = x + gn_slice_base(q, g, slice);
    const float* cb = q.chan_bias ? q.chan_bias + ((g / q.G) * q.slices + slice) * (int64_t)(q.Cg * q.G) + (g % q.G) * q.Cg : nullptr;
    const int64_t n_chunk = (E - e0) < CH ? (E - e0) : CH;
    const float inv_S = 1.0f / (float)q.S;
    if (VEC) {
        uint4 raw[kGnVecPerThread];
        float addv[kGnVecPerThread];
#pragma unroll
        for (int i = 0; i < kGnVecPerThread; ++i) {
            const int64_t e = e0 + ((int64_t)i * kGnBlock + threadIdx.x) * KV;
            const bool ok = e < E;
            raw[i] = ok ? *reinterpret_cast<const uint4*>(base + e) : make_uint4(0, 0, 0, 0);
            addv[i] = (ok && cb) ? cb[q.E < (1 << 24) ? div_small((uint32_t)e, (uint32_t)q.S, inv_S) : (int)(e / q.S)] : 0.f;
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < kGnVecPerThread; ++i) {
            if (e0 + ((int64_t)i * kGnBlock + threadIdx.x) * KV < E) {
                float t[KV];
                Io<T>::load(reinterpret_cast<const T*>(&raw[i]), t);
#pragma unroll
                for (int k = 0; k < KV; ++k) sum += t[k] + addv[i];
            }
        }
        // (the registers pass through an empty asm: the second pass must decode again instead of keeping the first pass's 64 floats alive)
#pragma unroll
        for (int i = 0; i < kGnVecPerThread; ++i) asm volatile("" : "+v"(raw[i].x), "+v"(raw[i].y), "+v"(raw[i].z), "+v"(raw[i].w));
        const float total = block_sum(sum, s_red);
        const float mean = total / (float)n_chunk;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < kGnVecPerThread; ++i) {
            if (e0 + ((int64_t)i * kGnBlock + threadIdx.x) * KV < E) {
                float t[KV];
                Io<T>::load(reinterpret_cast<const T*>(&raw[i]), t);
                const float off = mean - addv[i];
#pragma unroll
                for (int k = 0; k < KV; ++k) { const float d = t[k] - off; m2 += d * d; }
            }
        }
        m2 = block_sum(m2, s_red);
        if (threadIdx.x == 0) {
            float* p = part + (g * chunks + blockIdx.x) * 3;   // blockIdx.x = slice * cps + chunk
            p[0] = (float)n_chunk; p[1] = mean; p[2] = m2;
        }
        return;
    }
    float v[kGnVecPerThread * KV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kGnVecPerThread; ++i) {
        int64_t e = e0 + ((int64_t)i * kGnBlock + threadIdx.x) * KV;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            bool ok = e + k < E;
            v[i * KV + k] = ok ? Io<T>::ld1(base + e + k) + (cb ? cb[(e + k) / q.S] : 0.f) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) sum += v[i * KV + k];
    }
    const float total = block_sum(sum, s_red);
    const float mean = total / (float)n_chunk;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < kGnVecPerThread; ++i) {
        int64_t e = e0 + ((int64_t)i * kGnBlock + threadIdx.x) * KV;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            float d = v[i * KV + k] - mean;
            if (e + k < E) m2 += d * d;
        }
    }
    m2 = block_sum(m2, s_red);
    if (threadIdx.x == 0) {
        float* p = part + (g * chunks + blockIdx.x) * 3;   // blockIdx.x = slice * cps + chunk
        p[0] = (float)n_chunk; p[1] = mean; p[2] = m2;
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(kGnBlock) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ weight,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ part, GnGeom q, float eps, int silu) {
    __shared__ float s_stat[2];
    constexpr int KV = Io<T>::kVec;
    constexpr int CH = kGnBlock * kGnVecPerThread * KV;
    const int64_t g = blockIdx.y;
    const int slice = blockIdx.x / q.cps, chunk = blockIdx.x % q.cps;
    const int64_t E = q.E, S = q.S;
    const int chunks = q.cps * q.slices, Cg = q.Cg, G = q.G;
    if (threadIdx.x < 64) {
        // merge the group's partials: lanes take them round-robin, then a butterfly of Chan merges
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int c = threadIdx.x; c < chunks; c += 64) {
            const float* p = part + (g * chunks + c) * 3;
            float nb = p[0], mb = p[1], m2b = p[2];
            float nt = n + nb, d = mb - mean;
            mean += d * (nb / nt);
            m2 += m2b + d * d * (n * nb / nt);
            n = nt;
        }
        for (int o = 32; o > 0; o >>= 1) {
            float nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), m2b = __shfl_xor(m2, o);
            float nt = n + nb;
            if (nt > 0.f) {
                float d = mb - mean;
                float new_mean = mean + d * (nb / nt);
                m2 = m2 + m2b + d * d * (n * nb / nt);
                mean = new_mean;
            }
            n = nt;
        }
        if (threadIdx.x == 0) { s_stat[0] = mean; s_stat[1] = rsqrtf(m2 / n + eps); }
    }
    __syncthreads();
    const float mean = s_stat[0], rstd = s_stat[1];
    const int c0 = (int)(g % G) * Cg;
    const int C = Cg * G;
    const int64_t e0 = (int64_t)chunk * CH;
    const int64_t sb = gn_slice_base(q, g, slice);
    const int64_t row = (g / G) * q.slices + slice;       // first-dimension index of x
    const T* xb = x + sb;
    const float* cb = q.chan_bias ? q.chan_bias + row * C : nullptr;
    // output addressing: plain = same as x; stack3 = rows of 3C channels
    T* yb = y + sb;
    T *y_self = nullptr, *y_next = nullptr, *y_prev = nullptr, *y_zero = nullptr;
    if (q.stack3) {
        const int64_t rs = 3 * (int64_t)C * S;            // row stride of the stacked tensor
        const int64_t co = (int64_t)c0 * S;               // this group's channel offset inside a C-block
        y_self = y + row * rs + (int64_t)C * S + co;                                  // tap 1 of frame t
        y_next = slice + 1 < q.slices ? y + (row + 1) * rs + co : nullptr;            // tap 0 of frame t+1
        y_prev = slice > 0 ? y + (row - 1) * rs + 2 * (int64_t)C * S + co : nullptr;  // tap 2 of frame t-1
        // the ends of the sequence see a zero frame: frame 0's tap 0 and frame T-1's tap 2 (written by this block
        // when it owns frame 0 / frame T-1; a one-frame video needs both, handled by the second pointer below)
        y_zero = slice == 0 ? y + row * rs + co : nullptr;
    }
    T* y_zero2 = (q.stack3 && slice == q.slices - 1) ? y + row * (3 * (int64_t)C * S) + 2 * (int64_t)C * S + (int64_t)c0 * S : nullptr;
    const float inv_S = 1.0f / (float)S;
    // VEC: all of the thread's loads first, as loaded (4 registers each) — with a load, its arithmetic and its stores per turn of
    // the loop a thread had one load in flight at a time (23 registers: the occupancy was there, the loads were not)
    uint4 raw[VEC ? kGnVecPerThread : 1];
    if (VEC) {
#pragma unroll
        for (int i = 0; i < kGnVecPerThread; ++i) {
            const int64_t e = e0 + ((int64_t)i * kGnBlock + threadIdx.x) * KV;
            raw[i] = e < E ? *reinterpret_cast<const uint4*>(xb + e) : make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < kGnVecPerThread; ++i) {
        int64_t e = e0 + ((int64_t)i * kGnBlock + threadIdx.x) * KV;
        if (e >= E) continue;
        float v[KV];
        if (VEC) {
            Io<T>::load(reinterpret_cast<const T*>(&raw[i]), v);
            const int cl = E < (1 << 24) ? div_small((uint32_t)e, (uint32_t)S, inv_S) : (int)(e / S);   // S % KV == 0 on this path: one channel per vector
            const int c = c0 + cl;
            const float add = cb ? cb[c] : 0.f;
            float w = weight[c] * rstd, b = bias[c] + (add - mean) * w;
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                float t = v[k] * w + b;
                v[k] = silu ? t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f)) : t;
            }
            if (!q.stack3) {
                Io<T>::store(yb + e, v);
            } else {
                Io<T>::store(y_self + e, v);
                if (y_next) Io<T>::store(y_next + e, v);
                if (y_prev) Io<T>::store(y_prev + e, v);
                if (y_zero || y_zero2) {
                    float z[KV];
#pragma unroll
                    for (int k = 0; k < KV; ++k) z[k] = 0.f;
                    if (y_zero) Io<T>::store(y_zero + e, z);
                    if (y_zero2) Io<T>::store(y_zero2 + e, z);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                if (e + k >= E) break;
                int c = c0 + (int)((e + k) / S);
                float t = (Io<T>::ld1(xb + e + k) + (cb ? cb[c] : 0.f) - mean) * rstd * weight[c] + bias[c];
                t = silu ? t / (1.0f + __expf(-t)) : t;
                if (!q.stack3) {
                    Io<T>::st1(yb + e + k, t);
                } else {
                    Io<T>::st1(y_self + e + k, t);
                    if (y_next) Io<T>::st1(y_next + e + k, t);
                    if (y_prev) Io<T>::st1(y_prev + e + k, t);
                    if (y_zero) Io<T>::st1(y_zero + e + k, 0.f);
                    if (y_zero2) Io<T>::st1(y_zero2 + e + k, 0.f);
                }
            }
        }
    }
}

// Token-major output: y[n, s, c] = act(GroupNorm(x)[n, c, s]) — "b c h w -> b (h w) c" fused into the apply pass, for the
// consumers that contract over channels (the transformer's proj_in Linear, attention.py:700-707; the channels-last
// 3x3 convolution of ResBlock). One block = a 64 (channels) x 64 (positions) tile through LDS; the tile's channels
// merge their groups' partials themselves (a handful per group), so no extra finalise launch exists.
constexpr int kGnTok = 64;                                // channels per tile
constexpr int kGnTokS = 128;                              // positions per tile
// The tile sits in LDS in the OUTPUT type (the rounding of the final store, done before the transpose instead of after it:
// same values) — 17 KB for bf16 / f16 instead of 33 KB of floats, so the 8 blocks a CU's wave slots allow are resident and a
// thread has four 16-byte loads in flight instead of two (the 64 x 64 float tile ran at 4.2 TB/s of read + write).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_tokens_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                              const float* __restrict__ weight,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ part, GnGeom q, float eps, int silu,
                                                              int s_tiles, int c_tiles) {
    constexpr int V = Io<T>::kVec;
    constexpr int VPRS = kGnTokS / V;                     // vectors per channel row of the tile (load side)
    constexpr int VPRC = kGnTok / V;                      // vectors per token row of the tile (store side)
    constexpr int kPad = 16 / (int)sizeof(T);             // row pitch 128 + 16 bytes' worth: 16-byte aligned rows, transposed reads conflict-free
    __shared__ __attribute__((aligned(16))) T s_t[kGnTok][kGnTokS + kPad];
    __shared__ float s_sc[kGnTok], s_sh[kGnTok];
    int bid = blockIdx.x;
    const int stile = bid % s_tiles; bid /= s_tiles;
    const int ctile = bid % c_tiles;
    const int64_t n = bid / c_tiles;
    const int C = q.Cg * q.G;
    const int c0 = ctile * kGnTok;
    const int64_t s0 = (int64_t)stile * kGnTokS, S = q.S;
    if (threadIdx.x < kGnTok) {
        const int c = c0 + threadIdx.x;
        if (c < C) {
            const int64_t g = n * q.G + c / q.Cg;
            float cnt = 0.f, mean = 0.f, m2 = 0.f;
            // the partials four at a time, loads first: one at a time this walk was a chain of 6 ... 12 dependent L2 round trips in
            // front of every tile (a block's data phase is shorter than that)
            for (int k0 = 0; k0 < q.cps; k0 += 4) {
                float nb[4], mb[4], qb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* p = part + (g * q.cps + (k0 + u < q.cps ? k0 + u : q.cps - 1)) * 3;
                    nb[u] = k0 + u < q.cps ? p[0] : 0.f; mb[u] = p[1]; qb[u] = p[2];
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
            const float rstd = rsqrtf(m2 / cnt + eps);
            const float add = q.chan_bias ? q.chan_bias[n * C + c] : 0.f;
            const float w = weight[c] * rstd;
            s_sc[threadIdx.x] = w;
            s_sh[threadIdx.x] = bias[c] + (add - mean) * w;
        }
    }
    // the tile's loads do not depend on the scale / shift: issue them before the barrier
    constexpr int kLoads = kGnTok * VPRS / 256;
    uint4 raw[kLoads];
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
        const int i = threadIdx.x + 256 * j;
        const int cr = i / VPRS, sv = i % VPRS;
        raw[j] = (c0 + cr < C && s0 + sv * V < S) ? *reinterpret_cast<const uint4*>(x + ((n * C + c0 + cr) * S + s0 + sv * V)) : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
        const int i = threadIdx.x + 256 * j;
        const int cr = i / VPRS, sv = i % VPRS;
        float v[V];
        Io<T>::load(reinterpret_cast<const T*>(&raw[j]), v);
        const float w = s_sc[cr], b = s_sh[cr];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float t = v[k] * w + b;
            v[k] = silu ? t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f)) : t;
        }
        Io<T>::store(&s_t[cr][sv * V], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kGnTokS * VPRC; i += 256) {
        const int sr = i / VPRC, cv = i % VPRC;
        if (s0 + sr < S && c0 + cv * V < C) {
            T o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = s_t[cv * V + k][sr];
            *reinterpret_cast<uint4*>(y + ((n * S + s0 + sr) * C + c0 + cv * V)) = *reinterpret_cast<const uint4*>(o);
        }
    }
}

// Register-resident single pass: ONE 512-thread block per (sample, group) keeps the whole group in registers
// (NV 16-byte vectors per thread), so x is read from HBM exactly once: sum -> mean, exact centred second moment from the
// registers, normalise + affine (+ SiLU), store. The two-launch form above reads x twice (1.5x the algorithmic traffic:
// 165 MB activations do not stay in the 256 MB Infinity Cache between the launches once the output stream passes
// through it) and reached 0.39 of the HBM peak in the 14 x 576x1024 step; a group of that step is 11 KB ... 368 KB,
// against 512 KB of vector registers per CU. Groups that do not fit (more than 256 KB in bf16 / f16, 192 KB in fp32), the temporal form and the token-major /
// stacked outputs keep the two-launch kernels.
template <typename T, int NV, int kGnResBlock>
__global__ __launch_bounds__(kGnResBlock) void gn_resident_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ chan_bias, int Cg, int G,
                                                                  int64_t S, float eps, int silu) {
    constexpr int KV = Io<T>::kVec;
    __shared__ float s_red[kGnResBlock / 64];
    const int64_t g = blockIdx.x;
    const int64_t n = g / G;
    const int c0 = (int)(g % G) * Cg, C = Cg * G;
    const int64_t E = (int64_t)Cg * S;
    const int nvec = (int)(E / KV);
    const T* xb = x + (n * C + c0) * S;
    T* yb = y + (n * C + c0) * S;
    const float* cb = chan_bias ? chan_bias + n * C + c0 : nullptr;
    auto block_total = [&](float v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kGnResBlock / 64; ++w) t += s_red[w];
        return t;
    };
    // every load is issued unconditionally (index clamped to the group's last vector) so that all NV of them are in
    // flight together; a predicated load compiles to a branch with s_waitcnt vmcnt(0) behind it, i.e. NV serial round
    // trips to HBM (measured: 180 us instead of 118 for the two-launch form at 184 KB per group)
    uint4 r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * kGnResBlock + threadIdx.x;
        r[i] = *reinterpret_cast<const uint4*>(xb + (int64_t)(vi < nvec ? vi : nvec - 1) * KV);
    }
    const float inv_S = 1.0f / (float)S;
    auto chan_of = [&](int vi) { return div_small((uint32_t)(vi < nvec ? vi : nvec - 1) * KV, (uint32_t)S, inv_S); };   // E < 2^24 (host check)
    // the vector's channel bias (one channel per vector: S % KV == 0), re-read in every pass (a handful of cached floats)
    auto chan_add = [&](int vi) { return cb ? cb[chan_of(vi)] : 0.f; };
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float v[KV];
        Io<T>::load(reinterpret_cast<const T*>(&r[i]), v);
        const int vi = i * kGnResBlock + threadIdx.x;
        const float a = chan_add(vi);
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) part += v[k] + a;
        sum += vi < nvec ? part : 0.f;
    }
    const float mean = block_total(sum) / (float)E;
    // the group stays in registers in its STORAGE type: every pass converts again from r[] (a few VALU ops per vector;
    // without the opaque barrier the compiler keeps the fp32 copies of the first pass alive: 3x the registers)
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(r[i].x), "+v"(r[i].y), "+v"(r[i].z), "+v"(r[i].w));
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float v[KV];
        Io<T>::load(reinterpret_cast<const T*>(&r[i]), v);
        const int vi = i * kGnResBlock + threadIdx.x;
        const float a = chan_add(vi) - mean;
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) { const float d = v[k] + a; part += d * d; }
        m2 += vi < nvec ? part : 0.f;
    }
    const float rstd = rsqrtf(block_total(m2) / (float)E + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(r[i].x), "+v"(r[i].y), "+v"(r[i].z), "+v"(r[i].w));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * kGnResBlock + threadIdx.x;
        float v[KV];
        Io<T>::load(reinterpret_cast<const T*>(&r[i]), v);
        const int cl = chan_of(vi);
        const int c = c0 + cl;
        const float w = weight[c] * rstd, b = bias[c] + ((cb ? cb[cl] : 0.f) - mean) * w;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const float t = v[k] * w + b;
            // t * sigmoid(t) with the hardware exp2 / rcp (1 ulp each): 5 instructions instead of the ~15 of expf + IEEE division
            v[k] = silu ? t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f)) : t;
        }
        if (vi < nvec) Io<T>::store(yb + (int64_t)vi * KV, v);
    }
}

// (block size, NV): 512-thread blocks for groups of up to 12 vectors per thread (<= 96 registers: 3-4 such blocks per
// CU); larger groups take 1024-thread blocks so that the CU that owns one still has 4 waves per SIMD to hide the
// dependent-issue latency of the three register passes (at 2 waves per SIMD the 184 KB groups of level 0 took 120 us
// against 115 for the two-launch form)
template <typename T, int BS>
static int gn_resident_launch_bs(const void* x, void* y, const float* w, const float* b, const float* chan_bias, int64_t N, int C,
                                 int64_t S, int G, float eps, int silu, hipStream_t st, int nv) {
    const int Cg = C / G;
    const int64_t blocks = N * G;
#define MVI_GN_RES(NVV)                                                                                                          \
    case NVV:                                                                                                                    \
        hipLaunchKernelGGL((gn_resident_kernel<T, NVV, BS>), dim3((unsigned)blocks), dim3(BS), 0, st, (const T*)x, (T*)y, w, b, chan_bias, \
                           Cg, G, S, eps, silu);                                                                                  \
        break;
    switch (nv) {
        MVI_GN_RES(2) MVI_GN_RES(4) MVI_GN_RES(8) MVI_GN_RES(12)
        default: return 1;
    }
#undef MVI_GN_RES
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
template <typename T>
static int gn_resident_launch(const void* x, void* y, const float* w, const float* b, const float* chan_bias, int64_t N, int C,
                              int64_t S, int G, float eps, int silu, hipStream_t st) {
    constexpr int KV = Io<T>::kVec;
    const int64_t nvec = (int64_t)(C / G) * S / KV;
    if (N * G > 0x7FFFFFFFll || nvec >= (1 << 21)) return 1;      // caller falls back to the two-launch form
    const int opts[] = {2, 4, 8, 12};
    for (int bs : {512, 1024}) {
        const int64_t need = (nvec + bs - 1) / bs;
        for (int o : opts)
            if (need <= o)
                return bs == 512 ? gn_resident_launch_bs<T, 512>(x, y, w, b, chan_bias, N, C, S, G, eps, silu, st, o)
                                 : gn_resident_launch_bs<T, 1024>(x, y, w, b, chan_bias, N, C, S, G, eps, silu, st, o);
    }
    return 1;                                                      // more than 12 vectors per thread of a 1024-thread block
}

// Cluster form: ONE launch, x read once, for plain groups that do not fit one block's registers — the 270 ... 552 KB groups
// of the level-0 / level-1 decoder (the kernel also covers the temporal and stacked forms; gn_cluster_launch says why
// they are not routed here). A group is handled by K = slices * kps blocks; block j keeps part j % kps of slice j / kps in
// registers (NV 16-byte vectors per thread), computes its count / mean / centred M2 exactly (two passes over registers),
// publishes the triple and meets the other K - 1 blocks at a per-group counter; every block then merges the K triples
// in the same order (Chan) and normalises + stores its part. Two 512-thread blocks fit a CU (<= 128 registers), so one
// block's wait and store phases run under the other's loads.
//   * Exchange through memory: the triples and counters move with agent-scope relaxed atomics (the XCDs' L2s are not
//     coherent with each other); each thread drains its stores before the block arrives.
//   * Progress, by construction (round 3; round 2 relied on the observed workgroup-id -> XCD dispatch order): a block's place
//     (group, part) comes from a TICKET it draws when it starts (one atomic add), not from blockIdx. Tickets are handed out in
//     start order, so at any time at most ONE group is incomplete — the one the next ticket belongs to — and every other
//     waiting block belongs to a group whose K <= 8 blocks have all started: those groups finish without help, their slots
//     free, and the blocks that start next draw exactly the missing tickets. No block ever waits for a block that cannot
//     start, whatever the dispatcher does (other tenants, CU masks, partition modes). Blocks with consecutive tickets started
//     together, so a group's blocks also meet quickly. The spin stays bounded as a defence against a wedged device
//     (kGnSpinLimit, then sync[2 * kGnSyncGroups] is set); hip_ops.groupnorm_cluster_timeouts() reads that flag and the
//     engine / benchmark raise on it at the end of every sample / run.
//   * sync = 2 words per group [arrived, departed] + [timeout flag, ticket, departed blocks], zero before the launch and
//     zero again after it: the last block of a group re-arms the group's words, the last block of the launch the ticket. The
//     caller provides one zeroed buffer per (device, stream) (mvi_groupnorm_sync_bytes()).
constexpr int kGnClusterBlock = 512;
constexpr int kGnSyncGroups = 65536;
constexpr uint32_t kGnSpinLimit = 1u << 22;

template <typename T, int NV>
__global__ __launch_bounds__(kGnClusterBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void gn_cluster_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                     const float* __restrict__ weight,
                                                                     const float* __restrict__ bias, GnGeom q, int kps, int vpb,
                                                                     float eps, int silu, float* part, uint32_t* sync,
                                                                     int64_t groups) {
    constexpr int KV = Io<T>::kVec, BS = kGnClusterBlock;
    __shared__ float s_red[BS / 64];
    __shared__ float s_stat[2];
    const int K = q.slices * kps;
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(sync + 2 * kGnSyncGroups + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int64_t seq = s_ticket;                                        // place in START order (see above), not blockIdx
    const int64_t g = seq / K;
    const int j = (int)(seq % K), slice = j / kps, piece = j % kps;
    const int Cg = q.Cg, G = q.G, C = Cg * G;
    const int64_t S = q.S;
    const int nvec = (int)(q.E / KV);
    const int v0 = piece * vpb, v1 = min(nvec, v0 + vpb);               // this block's vectors of the slice (v1 > v0: host)
    const int64_t sb = gn_slice_base(q, g, slice);
    const int64_t row = (g / G) * q.slices + slice;
    const int c0 = (int)(g % G) * Cg;
    const T* xb = x + sb;
    const float* cb = q.chan_bias ? q.chan_bias + row * C + c0 : nullptr;
    auto block_total = [&](float v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < BS / 64; ++w) t += s_red[w];
        return t;
    };
    uint4 r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {                                       // unconditional, clamped: all NV loads in flight together
        const int vi = v0 + i * BS + threadIdx.x;
        r[i] = *reinterpret_cast<const uint4*>(xb + (int64_t)(vi < v1 ? vi : v1 - 1) * KV);
    }
    const float inv_S = 1.0f / (float)S;
    auto chan_of = [&](int vi) { return div_small((uint32_t)(vi < v1 ? vi : v1 - 1) * KV, (uint32_t)S, inv_S); };   // E < 2^24 (host)
    auto chan_add = [&](int vi) { return cb ? cb[chan_of(vi)] : 0.f; };
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float v[KV];
        Io<T>::load(reinterpret_cast<const T*>(&r[i]), v);
        const int vi = v0 + i * BS + threadIdx.x;
        const float a = chan_add(vi);
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) p += v[k] + a;
        sum += vi < v1 ? p : 0.f;
    }
    const float cnt_l = (float)(v1 - v0) * (float)KV;
    const float mean_l = block_total(sum) / cnt_l;
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(r[i].x), "+v"(r[i].y), "+v"(r[i].z), "+v"(r[i].w));
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float v[KV];
        Io<T>::load(reinterpret_cast<const T*>(&r[i]), v);
        const int vi = v0 + i * BS + threadIdx.x;
        const float a = chan_add(vi) - mean_l;
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) { const float d = v[k] + a; p += d * d; }
        m2 += vi < v1 ? p : 0.f;
    }
    const float m2_l = block_total(m2);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(r[i].x), "+v"(r[i].y), "+v"(r[i].z), "+v"(r[i].w));
    // publish, meet the group's other blocks, merge
    float* trip = part + (g * K) * 3;
    uint32_t* gs = sync + 2 * g;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(trip + 3 * j + 0, cnt_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(trip + 3 * j + 1, mean_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(trip + 3 * j + 2, m2_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(gs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0;
            while (__hip_atomic_load(gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)K) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kGnSpinLimit) {
                    __hip_atomic_store(sync + 2 * kGnSyncGroups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        asm volatile("" ::: "memory");                  // the triples are read after lane 0 left the spin (program order of the wave)
        // lane i holds block i's triple; lane-order Chan merge, the same in every block of the group
        const int li = threadIdx.x < K ? threadIdx.x : 0;
        const float nb_i = __hip_atomic_load(trip + 3 * li + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float mb_i = __hip_atomic_load(trip + 3 * li + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float qb_i = __hip_atomic_load(trip + 3 * li + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float n = 0.f, mean = 0.f, M2 = 0.f;
        for (int i = 0; i < K; ++i) {
            const float nb = __shfl(nb_i, i), mb = __shfl(mb_i, i), qb = __shfl(qb_i, i);
            const float nt = n + nb, d = mb - mean;
            mean += d * (nb / nt);
            M2 += qb + d * d * (n * nb / nt);
            n = nt;
        }
        if (threadIdx.x == 0) { s_stat[0] = mean; s_stat[1] = rsqrtf(M2 / n + eps); }
    }
    __syncthreads();
    const float mean = s_stat[0], rstd = s_stat[1];
    // output addressing (as gn_apply_kernel): plain = x's layout; stack3 = rows of 3C channels, taps self / next / prev + zero ends
    T* yb = y + sb;
    T *y_self = nullptr, *y_next = nullptr, *y_prev = nullptr, *y_zero = nullptr, *y_zero2 = nullptr;
    if (q.stack3) {
        const int64_t rs = 3 * (int64_t)C * S, co = (int64_t)c0 * S;
        y_self = y + row * rs + (int64_t)C * S + co;
        y_next = slice + 1 < q.slices ? y + (row + 1) * rs + co : nullptr;
        y_prev = slice > 0 ? y + (row - 1) * rs + 2 * (int64_t)C * S + co : nullptr;
        y_zero = slice == 0 ? y + row * rs + co : nullptr;
        y_zero2 = slice == q.slices - 1 ? y + row * rs + 2 * (int64_t)C * S + co : nullptr;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = v0 + i * BS + threadIdx.x;
        float v[KV];
        Io<T>::load(reinterpret_cast<const T*>(&r[i]), v);
        const int cl = chan_of(vi);
        const int c = c0 + cl;
        const float w = weight[c] * rstd, b = bias[c] + ((cb ? cb[cl] : 0.f) - mean) * w;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const float t = v[k] * w + b;
            v[k] = silu ? t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f)) : t;
        }
        if (vi < v1) {
            const int64_t e = (int64_t)vi * KV;
            if (!q.stack3) {
                Io<T>::store(yb + e, v);
            } else {
                Io<T>::store(y_self + e, v);
                if (y_next) Io<T>::store(y_next + e, v);
                if (y_prev) Io<T>::store(y_prev + e, v);
                if (y_zero || y_zero2) {
                    float z[KV];
#pragma unroll
                    for (int k = 0; k < KV; ++k) z[k] = 0.f;
                    if (y_zero) Io<T>::store(y_zero + e, z);
                    if (y_zero2) Io<T>::store(y_zero2 + e, z);
                }
            }
        }
    }
    // the last block to leave re-arms the group's counters for the next launch
    if (threadIdx.x == 0) {
        const uint32_t left = __hip_atomic_fetch_add(gs + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (uint32_t)K - 1u) {
            __hip_atomic_store(gs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gs + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ... and the last block of the launch the ticket counter (every block has drawn its ticket by then)
        uint32_t* all = sync + 2 * kGnSyncGroups + 2;
        if (__hip_atomic_fetch_add(all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(all, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 2 * kGnSyncGroups + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// returns 1 when the shape does not take the cluster form (caller falls through to the two-launch kernels)
template <typename T>
static int gn_cluster_launch(const void* x, void* y, const float* w, const float* b, const GnGeom& q, int64_t groups, float eps,
                             int silu, float* part, uint32_t* sync, hipStream_t st) {
    constexpr int KV = Io<T>::kVec, BS = kGnClusterBlock;
    if (q.E % KV != 0 || q.E >= (1 << 24) || groups > kGnSyncGroups) return 1;
    const int64_t nvec = q.E / KV;
    const int kps = (int)((nvec + 12 * BS - 1) / (12 * BS));
    const int64_t K = (int64_t)q.slices * kps;
    // the triples reuse the two-launch form's partial-statistics area: K <= slices * cps (a chunk is 2048 vectors, a block
    // holds up to 6144)
    // K <= 8 and one slice: measured on MI355X (tools/bench_groupnorm.py, tools/gn_census.py, bf16, us per call, two launches ->
    // cluster): plain groups of 360 KB (28, 640, 72, 128) 250 -> 200, 270 KB (28, 1920, 36, 64) 172 -> 148, 184 KB as two
    // pieces 110 (resident) -> 104; but the temporal groups LOSE — K = 28 (2.5 MB, stacked output) 165 -> 214, K = 14
    // (1.26 MB) 90 -> 87, (630 KB) 54 -> 60, (158 KB) 22 -> 41: a block waits for the slowest of its K - 1 partners, an XCD
    // starts the second block of a CU ~10 us after the first (tools/experiments/xcd_probe.cpp), and above 32 blocks per
    // group the wait did not end at all (K = 53: the bounded spin ran out) — those shapes keep the two launches.
    if (q.slices != 1 || K > 8 || K > (int64_t)q.cps || (groups + 8) * K > 0x7FFFFFFFll) return 1;
    const int vpb = (int)((nvec + kps - 1) / kps);
    if ((int64_t)vpb * (kps - 1) >= nvec) return 1;                      // every block must own at least one vector
    const int need = (vpb + BS - 1) / BS;
    const unsigned blocks = (unsigned)(groups * K);                       // exactly the tickets 0 .. groups * K - 1
#define MVI_GN_CL(NVV)                                                                                                        \
    hipLaunchKernelGGL((gn_cluster_kernel<T, NVV>), dim3(blocks), dim3(BS), 0, st, (const T*)x, (T*)y, w, b, q, kps, vpb, eps, silu, \
                       part, sync, groups)
    if (need <= 2) MVI_GN_CL(2);
    else if (need <= 4) MVI_GN_CL(4);
    else if (need <= 8) MVI_GN_CL(8);
    else MVI_GN_CL(12);
#undef MVI_GN_CL
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

template <typename T>
static int gn_launch(const void* x, void* y, const float* w, const float* b, const float* chan_bias, int stack3, int64_t N,
                     int slices, int C, int64_t S, int G, float eps, int silu, float* part, hipStream_t st, int tokens = 0,
                     uint32_t* sync = nullptr) {
    constexpr int KV = Io<T>::kVec;
    constexpr int CH = kGnBlock * kGnVecPerThread * KV;
    GnGeom q;
    q.Cg = C / G; q.G = G; q.S = S; q.slices = slices;
    q.E = (int64_t)q.Cg * S;
    q.slice_stride = (int64_t)C * S;
    q.cps = (int)((q.E + CH - 1) / CH);
    q.chan_bias = chan_bias; q.stack3 = stack3;
    const bool vec = (S % KV == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    dim3 grid((unsigned)(q.cps * slices), (unsigned)(N * G));
    if (tokens) {
        if (!vec || C % KV != 0 || slices != 1 || stack3) return MVI_EINVAL;
        const int s_tiles = (int)((S + kGnTokS - 1) / kGnTokS), c_tiles = (C + kGnTok - 1) / kGnTok;
        const int64_t blocks = N * s_tiles * c_tiles;
        if (blocks > 0x7FFFFFFFll) return MVI_EINVAL;
        hipLaunchKernelGGL((gn_stats_kernel<T, true>), grid, dim3(kGnBlock), 0, st, (const T*)x, part, q);
        hipLaunchKernelGGL((gn_apply_tokens_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)y, w, b, part,
                           q, eps, silu, s_tiles, c_tiles);
        return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
    }
    // measured (tools/bench_groupnorm.py, bf16, MI355X, two launches -> resident): 11 KB groups 21.7 -> 17.8 us, 46 KB
    // 40.4 -> 27.5, 92 KB 59.0 -> 46.3 and 55.9 -> 44.3, 184 KB (one 1024-thread block per CU: load, three register
    // passes and store run in series there) 113.8 -> 109.2; larger groups do not fit and keep the two launches
    static const int resident_kb = getenv("MVI_GN_RESIDENT_KB") ? atoi(getenv("MVI_GN_RESIDENT_KB")) : 192;
    // (the resident kernel indexes channels with 24-bit arithmetic: groups of up to 256 KB are far below that)
    if (vec && slices == 1 && !stack3 && (int64_t)q.Cg * S * (int64_t)sizeof(T) <= (int64_t)resident_kb * 1024) {
        const int rc = gn_resident_launch<T>(x, y, w, b, chan_bias, N, C, S, G, eps, silu, st);
        if (rc != 1) return rc;
    }
    // groups from cluster_kb up (and every temporal / stacked call) take the cluster form when the caller gave a sync buffer
    static const int cluster_kb = getenv("MVI_GN_CLUSTER_KB") ? atoi(getenv("MVI_GN_CLUSTER_KB")) : 0;
    if (vec && sync && cluster_kb >= 0 && q.E * slices * (int64_t)sizeof(T) >= (int64_t)cluster_kb * 1024) {
        const int rc = gn_cluster_launch<T>(x, y, w, b, q, N * G, eps, silu, part, sync, st);
        if (rc != 1) return rc;
    }
    // (Measured and not kept: running the two launches per BAND of samples, sized so that the statistics pass's reads are
    // still in the 256 MB Infinity Cache when the apply pass reads them again — 48 MB bands: 146 us instead of 116 at
    // (28, 320, 72, 128) bf16; smaller bands worse: the shorter grids cost more than the cached re-read saves.)
    if (vec) {
        hipLaunchKernelGGL((gn_stats_kernel<T, true>), grid, dim3(kGnBlock), 0, st, (const T*)x, part, q);
        hipLaunchKernelGGL((gn_apply_kernel<T, true>), grid, dim3(kGnBlock), 0, st, (const T*)x, (T*)y, w, b, part, q, eps, silu);
    } else {
        hipLaunchKernelGGL((gn_stats_kernel<T, false>), grid, dim3(kGnBlock), 0, st, (const T*)x, part, q);
        hipLaunchKernelGGL((gn_apply_kernel<T, false>), grid, dim3(kGnBlock), 0, st, (const T*)x, (T*)y, w, b, part, q, eps, silu);
    }
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi

static thread_local char g_uerr[384] = "";
extern "C" const char* mvi_unet_last_error(void) { return g_uerr; }
namespace mvi {
int unet_fail(int code, const char* msg) { snprintf(g_uerr, sizeof(g_uerr), "%s", msg); return code; }
}

static int chunks_for(int64_t E, int dtype) {
    int kv = dtype == MVI_DT_F32 ? 4 : 8;
    int64_t ch = (int64_t)mvi::kGnBlock * mvi::kGnVecPerThread * kv;
    return (int)((E + ch - 1) / ch);
}

extern "C" size_t mvi_groupnorm_workspace_bytes(int64_t N, int32_t C, int64_t spatial, int32_t groups) {
    if (N <= 0 || C <= 0 || groups <= 0 || spatial <= 0) return 0;
    int64_t E = (int64_t)(C / groups) * spatial;
    // sized for the dtype with the fewest elements per chunk (fp32); N counts every (video, frame) row, so the
    // same size covers the temporal form (N/T groups x T slices)
    return (size_t)(N * groups) * chunks_for(E, MVI_DT_F32) * 3 * sizeof(float);
}

static int gn_dispatch(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias, int stack3,
                       int64_t Nv, int32_t T, int32_t C,
                       int64_t spatial, int32_t groups, float eps, int32_t fuse_silu, int32_t dtype, void* workspace,
                       size_t workspace_bytes, void* stream, int tokens = 0, uint32_t* sync = nullptr) {
    if (Nv < 0 || T <= 0 || C <= 0 || groups <= 0 || spatial < 0 || C % groups != 0)
        return mvi::unet_fail(MVI_EINVAL, "groupnorm: C must be a positive multiple of groups");
    if (Nv == 0 || spatial == 0) return MVI_OK;
    if (!x || !y || !weight || !bias || !workspace) return mvi::unet_fail(MVI_EINVAL, "groupnorm: NULL pointer");
    if (Nv * groups > 65535) return mvi::unet_fail(MVI_EINVAL, "groupnorm: N*groups exceeds grid.y (65535)");
    if (workspace_bytes < mvi_groupnorm_workspace_bytes(Nv * T, C, spatial, groups))
        return mvi::unet_fail(MVI_ENOMEM, "groupnorm: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    int rc;
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::gn_launch<float>(x, y, weight, bias, chan_bias, stack3, Nv, T, C, spatial, groups, eps, fuse_silu, part, st, tokens, sync); break;
        case MVI_DT_BF16: rc = mvi::gn_launch<__hip_bfloat16>(x, y, weight, bias, chan_bias, stack3, Nv, T, C, spatial, groups, eps, fuse_silu, part, st, tokens, sync); break;
        case MVI_DT_F16: rc = mvi::gn_launch<__half>(x, y, weight, bias, chan_bias, stack3, Nv, T, C, spatial, groups, eps, fuse_silu, part, st, tokens, sync); break;
        default: return mvi::unet_fail(MVI_EINVAL, "groupnorm: unknown dtype");
    }
    if (rc == MVI_EINVAL)
        return mvi::unet_fail(MVI_EINVAL, "groupnorm (token-major output): C and spatial must be multiples of the 16-byte vector, 16-B aligned");
    return rc ? mvi::unet_fail(MVI_EHIP, "groupnorm: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_groupnorm_silu(const void* x, void* y, const float* weight, const float* bias, int64_t N,
                                  int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu,
                                  int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return gn_dispatch(x, y, weight, bias, nullptr, 0, N, 1, C, spatial, groups, eps, fuse_silu, dtype, workspace, workspace_bytes, stream);
}

extern "C" int mvi_groupnorm_silu_temporal(const void* x, void* y, const float* weight, const float* bias,
                                           int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups,
                                           float eps, int32_t fuse_silu, int32_t dtype, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    return gn_dispatch(x, y, weight, bias, nullptr, 0, videos, T, C, spatial, groups, eps, fuse_silu, dtype, workspace, workspace_bytes, stream);
}

extern "C" int mvi_groupnorm_silu_ex(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                     int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups, float eps,
                                     int32_t fuse_silu, int32_t stack3, int32_t dtype, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (stack3 && x == y) return mvi::unet_fail(MVI_EINVAL, "groupnorm: stack3 output cannot alias the input");
    return gn_dispatch(x, y, weight, bias, chan_bias, stack3 ? 1 : 0, videos, T, C, spatial, groups, eps, fuse_silu, dtype, workspace, workspace_bytes, stream);
}

extern "C" size_t mvi_groupnorm_sync_bytes(void) { return (2 * (size_t)mvi::kGnSyncGroups + 3) * sizeof(uint32_t); }

extern "C" int mvi_groupnorm_silu_ex2(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                      int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups, float eps,
                                      int32_t fuse_silu, int32_t stack3, int32_t dtype, void* workspace,
                                      size_t workspace_bytes, void* sync, size_t sync_bytes, void* stream) {
    if (stack3 && x == y) return mvi::unet_fail(MVI_EINVAL, "groupnorm: stack3 output cannot alias the input");
    if (sync && sync_bytes < mvi_groupnorm_sync_bytes()) return mvi::unet_fail(MVI_ENOMEM, "groupnorm: sync buffer too small");
    return gn_dispatch(x, y, weight, bias, chan_bias, stack3 ? 1 : 0, videos, T, C, spatial, groups, eps, fuse_silu, dtype, workspace,
                       workspace_bytes, stream, 0, (uint32_t*)sync);
}

extern "C" int mvi_groupnorm_silu_tokens(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                         int64_t N, int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu,
                                         int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (x == y) return mvi::unet_fail(MVI_EINVAL, "groupnorm: token-major output cannot alias the input");
    return gn_dispatch(x, y, weight, bias, chan_bias, 0, N, 1, C, spatial, groups, eps, fuse_silu, dtype, workspace, workspace_bytes, stream, 1);
}
