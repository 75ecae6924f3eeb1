// bf16/f16 MFMA flash attention (forward) for gfx950, head dim 64 — the only dense contraction of
// the SVD denoise loop that is hand-written (BASELINE.json north_star). Replaces
// xformers.ops.memory_efficient_attention / F.scaled_dot_product_attention
// (svd_inpaint1/sgm/modules/attention.py:427-439, :332-336) for the spatial self-attention calls:
// (B*H, S) = (140, 9216), (280, 2304), (560, 576), (560, 144) at 14 x 576x1024.
//
// Layout: q/out [B, Sq, H, 64], k/v [B, Sk, H, 64] token-major, contiguous.
// Block = 4 wave64 = 128 query rows of one (batch, head); each wave owns 32 rows. KV tile = 64 keys.
// Per wave and KV tile (v_mfma_f32_32x32x16, fp32 accumulate):
//   S^T[key][query] = K Q^T    : 2 key blocks x 4 d-steps  = 8 MFMA   (A = K rows from LDS, B = Q in regs)
//   online softmax in registers: the query sits on the lane (col = lane & 31), its 64 scores are 2 x 16
//                                registers in this lane and in lane ^ 32 -> one cross-lane max, no LDS
//   O^T[d][query]  += V^T P^T  : 2 d blocks x 4 key-steps   = 8 MFMA   (A = V^T from LDS, B = P from the
//                                S^T accumulators, converted in place: MFMA C/D layout == next B layout
//                                up to the fixed key permutation the V^T fragment read follows)
// K tile rows are padded to 72 elements (b128 reads conflict-free), V is stored transposed with a
// row of 68 elements (b64 reads conflict-free).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int kFD = 64;          // head dim
constexpr int kFQ = 128;         // query rows per block
constexpr int kFK = 64;          // keys per tile
constexpr int kKStride = 72;     // elements per K row in LDS
constexpr int kVStride = 68;     // elements per V^T row in LDS

template <typename T> struct Mma;
template <> struct Mma<__hip_bfloat16> {
    using frag = bf16x8;
    __device__ static f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {     // one v_cvt_pk_bf16_f32 (RNE)
        f32x2 f = {lo, hi};
        bf16x2 r = __builtin_convertvector(f, bf16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
};
template <> struct Mma<__half> {
    using frag = f16x8;
    __device__ static f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        f16x2 r = __builtin_convertvector(f, f16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
};

template <typename F> __device__ __forceinline__ F as_frag(u32x4 v) { return *reinterpret_cast<F*>(&v); }
template <typename T> __device__ __forceinline__ constexpr uint32_t kOnes2();                      // two 1.0 in the I/O type
template <> __device__ __forceinline__ constexpr uint32_t kOnes2<__hip_bfloat16>() { return 0x3F803F80u; }
template <> __device__ __forceinline__ constexpr uint32_t kOnes2<__half>() { return 0x3C003C00u; }

// Build variants kept for A/B runs (tools/ab_attention.sh; micro-benchmark B28 H5 S9216 / B28 H10 S2304, bf16, TFLOP/s):
//   PIPE=1 WPE=3 (default)            781 / 761     QK^T of tile t+1 issued under the softmax of tile t
//   PIPE=0 WPE=3                      768 / 738     one score accumulator, overlap left to the SIMD's other waves
//   PIPE=0 WPE=4                      783 / 744     ... at 4 waves per SIMD (128 VGPRs, 7 spilled)
//   PIPE=0 WPE=3 MFMA_ROWSUM=1        699 / 690     row sums as 4 extra MFMAs per tile instead of 33 v_add_f32
// The last line shows the matrix pipe is NOT idle enough to take 25 % more work: the loop is balanced between the two
// pipes at ~780 TFLOP/s, which is why removing VALU instructions alone (the first three lines) does not move it.
#ifndef MVI_ATTN_PIPE
#define MVI_ATTN_PIPE 1           // 1: QK^T of tile t+1 issued under the softmax of tile t (second score accumulator)
#endif
#ifndef MVI_ATTN_MFMA_ROWSUM
#define MVI_ATTN_MFMA_ROWSUM 0    // (non-pipelined variant only) softmax row sums as 4 extra MFMAs per tile
#endif
#ifndef MVI_ATTN_WPE
#define MVI_ATTN_WPE 3            // waves per SIMD the kernel is compiled for (register budget 512 / MVI_ATTN_WPE)
#endif

constexpr float kRescaleThreshold = 8.0f;   // log2 units: O and l are rescaled only when the row max grows by > 2^8

// Software pipeline: one wave keeps BOTH pipes busy. While the VALU runs the softmax of tile t (scores
// computed in the previous iteration), the matrix pipe runs S^T = K Q^T of tile t+1; the PV MFMAs of
// tile t follow as their P fragments come out of the converts (the compiler's own schedule of that
// single basic block spaces the 16 MFMAs ~8 VALU issues apart; sched_group_barrier hints made it worse).
// Measured and rejected: a 2x unrolled loop whose two score accumulators swap roles (removes the 16 v_mov_b64 of the
// loop latch) — 168 VGPRs + 43 spilled at 3 waves/SIMD: 597 TFLOP/s, 746 at 2 waves/SIMD, against 778 for this form.
// Costs a second score accumulator; 163 VGPRs, pinned to 3 waves/SIMD. The issue port, not the matrix
// pipe, bounds the loop at D = 64: 32 exp (8 cyc) + 32 fma + 32 add + 16 max3 + 16 cvt + 16 MFMA issue
// slots ~ 900 cycles per wave-tile against 512 matrix-pipe cycles. LDS holds K_{t+1} / K_{t+2} and V_t / V_{t+1}: two buffers each, one
// barrier per tile (K_{t+2} overwrites K_t, whose last reader finished before the previous barrier;
// V_{t+1} overwrites V_{t-1} likewise).
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MVI_ATTN_WPE, MVI_ATTN_WPE))) void attn_flash_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                   const T* __restrict__ v, T* __restrict__ out, int H,
                                                                   int Sq, int Sk, float scale_log2e, int q_blocks,
                                                                   int total_blocks, int64_t q_rs, int64_t kv_rs,
                                                                   int64_t o_rs) {
    using M = Mma<T>;
    using frag = typename M::frag;
    __shared__ __attribute__((aligned(16))) uint16_t s_k[2][kFK * kKStride];
    __shared__ __attribute__((aligned(16))) uint16_t s_vt[2][kFD * kVStride];

    int bid = blockIdx.x;
    if ((total_blocks & 7) == 0) bid = (bid & 7) * (total_blocks >> 3) + (bid >> 3);
    const int qb = bid % q_blocks;
    const int bh = bid / q_blocks;
    const int h = bh % H;
    const int64_t b = bh / H;
    // q_rs / kv_rs / o_rs: elements between consecutive tokens of q, of k and v, of out (H * 64 when contiguous;
    // 3 * H * 64 for q, k, v taken out of one packed projection)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int qcol = lane & 31, hh = lane >> 5;
    const int q0 = qb * kFQ + wave * 32;
    const int qrow = q0 + qcol;

    frag qf[4];
    {
        const T* qp = q + ((b * Sq + (qrow < Sq ? qrow : 0)) * q_rs + (int64_t)h * kFD + 8 * hh);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 raw = qrow < Sq ? *reinterpret_cast<const u32x4*>(qp + 16 * s) : u32x4{0, 0, 0, 0};
            qf[s] = as_frag<frag>(raw);
        }
    }
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m = -INFINITY, l = 0.f;

    const int kr = tid >> 2, ks = tid & 3;
    const int vp = tid >> 3, vs = tid & 7;
    const T* kbase = k + (b * Sk * kv_rs + (int64_t)h * kFD);
    const T* vbase = v + (b * Sk * kv_rs + (int64_t)h * kFD);
    u32x4 rk0, rk1, rv0, rv1;
    const u32x4 z4 = {0, 0, 0, 0};
    auto load_k = [&](int k0) {
        int r = k0 + kr;
        const T* p = kbase + (int64_t)r * kv_rs + 16 * ks;
        rk0 = r < Sk ? *reinterpret_cast<const u32x4*>(p) : z4;
        rk1 = r < Sk ? *reinterpret_cast<const u32x4*>(p + 8) : z4;
    };
    auto load_v = [&](int k0) {
        int r0 = k0 + 2 * vp;
        const T* pv = vbase + (int64_t)r0 * kv_rs + 8 * vs;
        rv0 = r0 < Sk ? *reinterpret_cast<const u32x4*>(pv) : z4;
        rv1 = r0 + 1 < Sk ? *reinterpret_cast<const u32x4*>(pv + kv_rs) : z4;
    };
    auto store_k = [&](int buf) {
        uint16_t* sk = s_k[buf];
        *reinterpret_cast<u32x4*>(&sk[kr * kKStride + 16 * ks]) = rk0;
        *reinterpret_cast<u32x4*>(&sk[kr * kKStride + 16 * ks + 8]) = rk1;
    };
    auto store_v = [&](int buf) {
        uint16_t* sv = s_vt[buf];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t a = rv0[i], c = rv1[i];
            uint32_t lo = (a & 0xFFFFu) | (c << 16), hi = (a >> 16) | (c & 0xFFFF0000u);
            *reinterpret_cast<uint32_t*>(&sv[(8 * vs + 2 * i) * kVStride + 2 * vp]) = lo;
            *reinterpret_cast<uint32_t*>(&sv[(8 * vs + 2 * i + 1) * kVStride + 2 * vp]) = hi;
        }
    };
    auto qk = [&](const uint16_t* sk, f32x16* st) {
        u32x4 kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                kf[kb][s] = *reinterpret_cast<const u32x4*>(&sk[(kb * 32 + qcol) * kKStride + 16 * s + 8 * hh]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) st[kb] = M::mfma(as_frag<frag>(kf[kb][s]), qf[s], st[kb]);
        }
    };

#if MVI_ATTN_PIPE
    const int n_tiles = (Sk + kFK - 1) / kFK;
    // prologue: K_0, V_0 -> buffers 0; K_1 -> buffer 1; scores of tile 0
    load_k(0); load_v(0);
    store_k(0); store_v(0);
    if (n_tiles > 1) { load_k(kFK); store_k(1); }
    __syncthreads();
    f32x16 st[2], sn[2];
    qk(s_k[0], st);

    auto tile = [&](int t, auto has_next_c) {
        constexpr bool kHasNext = decltype(has_next_c)::value;
        const int k0 = t * kFK;
        if (t + 2 < n_tiles) load_k((t + 2) * kFK);             // lands while this tile is processed
        if (kHasNext) load_v((t + 1) * kFK);
        // ---- row max of tile t, reference exponent (VALU only; rarely rescales)
        if (!kHasNext && k0 + kFK > Sk) {                       // only the last tile can be ragged
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((k0 + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) st[kb][r] = -INFINITY;
        }
        float rmax = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) rmax = fmaxf(rmax, fmaxf(st[0][r], st[1][r]));
        rmax = fmaxf(rmax, __shfl_xor(rmax, 32)) * scale_log2e;
        const bool grow = rmax > m + kRescaleThreshold;
        if (__ballot(grow) != 0ull) {
            const float m_new = grow ? rmax : m;
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
            m = m_new;
        }
        // ---- one scheduling region: QK^T of tile t+1 (matrix pipe) under exp/convert of tile t (VALU),
        //      then PV of tile t as its P fragments become available
        if (kHasNext) qk(s_k[(t + 1) & 1], sn);
        const uint16_t* sv = s_vt[t & 1];
        float rsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 pr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][8 * s2 + 2 * i], scale_log2e, -m));
                    float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][8 * s2 + 2 * i + 1], scale_log2e, -m));
                    rsum += p0 + p1;
                    pr[i] = M::pack2(p0, p1);
                }
                const frag pf = as_frag<frag>(pr);
                const int koff = 32 * kb + 16 * s2 + 4 * hh;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const uint16_t* row = &sv[(32 * db + qcol) * kVStride + koff];
                    u32x2 a0 = *reinterpret_cast<const u32x2*>(row);
                    u32x2 a1 = *reinterpret_cast<const u32x2*>(row + 8);
                    u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
                    o[db] = M::mfma(as_frag<frag>(av), pf, o[db]);
                }
            }
        l += rsum;
        if (t + 2 < n_tiles) store_k(t & 1);                     // K_t is dead: its scores exist since the last iteration
        if (kHasNext) store_v((t + 1) & 1);                      // V_{t-1} is dead
        if (kHasNext) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[0][i] = sn[0][i]; st[1][i] = sn[1][i]; }
        }
    };
    for (int t = 0; t + 1 < n_tiles; ++t) tile(t, std::true_type{});
    tile(n_tiles - 1, std::false_type{});
#else
    const int n_tiles = (Sk + kFK - 1) / kFK;
    // Variant without the intra-wave software pipeline (MVI_ATTN_PIPE=0): QK^T of tile t is computed at the start of
    // tile t into the ONE score accumulator; overlap of MFMA and VALU work comes from the other waves of the SIMD.
    load_k(0); load_v(0);
    store_k(0); store_v(0);
    __syncthreads();
    f32x16 st[2];
#if MVI_ATTN_MFMA_ROWSUM
    // Row sums on the matrix pipe: ones[32 x 16] . P^T[16 x 32] adds the 16 (rounded) probabilities of a k-step for every
    // query column into all 32 rows of an accumulator — 4 extra MFMAs per tile instead of 33 v_add_f32 on the issue
    // port that bounds the kernel. l is then the sum of exactly the values P.V uses.
    f32x16 lacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
    const frag ones = as_frag<frag>(u32x4{kOnes2<T>(), kOnes2<T>(), kOnes2<T>(), kOnes2<T>()});
#endif
    for (int t = 0; t < n_tiles; ++t) {
        const bool has_next = t + 1 < n_tiles;
        const int k0 = t * kFK;
        if (has_next) { load_k((t + 1) * kFK); load_v((t + 1) * kFK); }
        qk(s_k[t & 1], st);
        if (!has_next && k0 + kFK > Sk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((k0 + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) st[kb][r] = -INFINITY;
        }
        float rmax = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) rmax = fmaxf(rmax, fmaxf(st[0][r], st[1][r]));
        rmax = fmaxf(rmax, __shfl_xor(rmax, 32)) * scale_log2e;
        const bool grow = rmax > m + kRescaleThreshold;
        if (__ballot(grow) != 0ull) {
            const float m_new = grow ? rmax : m;
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
#if MVI_ATTN_MFMA_ROWSUM
#pragma unroll
            for (int i = 0; i < 16; ++i) lacc[i] *= alpha;
#endif
            m = m_new;
        }
        const uint16_t* sv = s_vt[t & 1];
        float rsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 pr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][8 * s2 + 2 * i], scale_log2e, -m));
                    float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][8 * s2 + 2 * i + 1], scale_log2e, -m));
#if !MVI_ATTN_MFMA_ROWSUM
                    rsum += p0 + p1;
#endif
                    pr[i] = M::pack2(p0, p1);
                }
                const frag pf = as_frag<frag>(pr);
#if MVI_ATTN_MFMA_ROWSUM
                lacc = M::mfma(ones, pf, lacc);
#endif
                const int koff = 32 * kb + 16 * s2 + 4 * hh;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const uint16_t* row = &sv[(32 * db + qcol) * kVStride + koff];
                    u32x2 a0 = *reinterpret_cast<const u32x2*>(row);
                    u32x2 a1 = *reinterpret_cast<const u32x2*>(row + 8);
                    u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
                    o[db] = M::mfma(as_frag<frag>(av), pf, o[db]);
                }
            }
        l += rsum;
        if (has_next) {
            store_k((t + 1) & 1);                                // buffers (t+1)&1 were last read in tile t-1
            store_v((t + 1) & 1);
            __syncthreads();
        }
    }
#if MVI_ATTN_MFMA_ROWSUM
    l = lacc[0];            // every row of the accumulator holds the column's sum (lane halves hold the same D element rows)
#endif
#endif
#if MVI_ATTN_PIPE || !MVI_ATTN_MFMA_ROWSUM
    l += __shfl_xor(l, 32);                                  // the two lane halves hold disjoint keys of every k-step
#endif
    if (qrow < Sq) {
        const float inv = 1.0f / l;
        T* op = out + ((b * Sq + qrow) * o_rs + (int64_t)h * kFD);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w = {M::pack2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                           M::pack2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)};
                *reinterpret_cast<u32x2*>(op + 32 * db + 8 * g + 4 * hh) = w;
            }
    }
}

// scale_log2e: what a score is multiplied by on its way into exp2 (softmax scale * log2 e; 1 for a q that carries it already)
template <typename T>
int attn_flash_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk,
                      float scale_log2e, hipStream_t st, int64_t q_rs, int64_t kv_rs, int64_t o_rs) {
    const int64_t hd = (int64_t)H * kFD;
    if (q_rs == 0) q_rs = hd;
    if (kv_rs == 0) kv_rs = hd;
    if (o_rs == 0) o_rs = hd;
    const int q_blocks = (Sq + kFQ - 1) / kFQ;
    const int64_t total = (int64_t)B * H * q_blocks;
    if (total > 0x7FFFFFFFll) return MVI_EINVAL;
    hipLaunchKernelGGL((attn_flash_kernel<T>), dim3((unsigned)total), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)v, (T*)out, H, Sq, Sk, scale_log2e, q_blocks, (int)total, q_rs, kv_rs, o_rs);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
template int attn_flash_launch<__hip_bfloat16>(const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t, int64_t, int64_t, int64_t);
template int attn_flash_launch<__half>(const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t, int64_t, int64_t, int64_t);

// 8-wave kernel for long sequences (attn_flash8.hip)
template <typename T>
int attn_flash8_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk,
                       float scale, bool q_log2, hipStream_t st, int64_t q_rs, int64_t kv_rs, int64_t o_rs);

// the same kernel on v_mfma_f32_16x16x32 (attn_flash8m16.hip; MVI_ATTN_MFMA16=1)
template <typename T>
int attn_flash8m16_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk,
                          float scale, bool q_log2, hipStream_t st, int64_t q_rs, int64_t kv_rs, int64_t o_rs);

// rowtile kernel (attn_rowtile.hip)
template <typename T>
int attn_rowtile_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk, int D,
                        float scale, hipStream_t st, int temporal_inner, int64_t q_ts, int64_t kv_ts, int64_t o_ts);
int unet_fail(int code, const char* msg);

}  // namespace mvi

#ifndef MVI_ATTN_MFMA16_DEFAULT
#define MVI_ATTN_MFMA16_DEFAULT 2      // 0: attn_flash8.hip (32x32x16); 1: 16x16x32, row sums on the VALU; 2: 16x16x32, row sums from the matrix pipe — profiles/round6_attention_mfma16_ab.txt: 2 is 2.7 - 3.0 % faster than 0 on both shapes, same box
#endif
extern "C" int mvi_attention_kernel_kind(int32_t Sq, int32_t Sk, int32_t D, int32_t dtype) {
    return (dtype != MVI_DT_F32 && D == mvi::kFD && Sk > 32) ? 1 : 0;
}
// the ONE place that picks the kernel: mvi_attention_forward* and the tests' assertion read the same answer
extern "C" int mvi_attention_kernel_variant(int32_t Sq, int32_t Sk, int32_t D, int32_t dtype) {
    if (mvi_attention_kernel_kind(Sq, Sk, D, dtype) != 1) return 0;
    // 256-row blocks pay off once there are enough of them and the padding of the last block is small
    static const int forced = getenv("MVI_ATTN_VARIANT") ? atoi(getenv("MVI_ATTN_VARIANT")) : 0;   // 4 / 8: force a kernel (A/B runs)
    // 16: the 8-wave kernel on v_mfma_f32_16x16x32 (attn_flash8m16.hip) wherever the 8-wave kernel would run
    static const int mfma16 = getenv("MVI_ATTN_MFMA16") ? atoi(getenv("MVI_ATTN_MFMA16")) : MVI_ATTN_MFMA16_DEFAULT;
    const bool eight = forced == 8 || (forced != 4 && Sq >= 1024 && Sk >= 256);
    return eight ? (mfma16 ? 16 : 8) : 4;
}

// q_log2: q carries softmax scale * log2(e) already (mvi_attention_forward_strided_qlog2); `scale` is then ln 2, what the kernels
// that exponentiate with e must apply, and the exp2 kernels take their scores as they are
constexpr float kLn2 = 0.6931471805599453f, kLog2e = 1.4426950408889634f;
static int attention_forward_impl(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t Sq,
                                  int32_t Sk, int32_t D, float scale, int32_t dtype, int64_t q_ts, int64_t kv_ts, int64_t o_ts,
                                  void* stream, bool q_log2 = false) {
    if (q_log2) scale = kLn2;
    if (B < 0 || H <= 0 || Sq < 0 || Sk <= 0 || D <= 0) return mvi::unet_fail(MVI_EINVAL, "attention: bad shape");
    if (B == 0 || Sq == 0) return MVI_OK;
    if (!q || !k || !v || !out) return mvi::unet_fail(MVI_EINVAL, "attention: NULL pointer");
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 != 0)
        return mvi::unet_fail(MVI_EINVAL, "attention: pointers must be 16-byte aligned");
    const int64_t hd = (int64_t)H * D;
    const int esz = dtype == MVI_DT_F32 ? 4 : 2;
    if (q_ts < 0 || kv_ts < 0 || o_ts < 0 || (q_ts && q_ts < hd) || (kv_ts && kv_ts < hd) || (o_ts && o_ts < hd) ||
        (q_ts * esz) % 16 || (kv_ts * esz) % 16 || (o_ts * esz) % 16)
        return mvi::unet_fail(MVI_EINVAL, "attention: token strides must be 0 or >= H*D elements and 16-byte multiples");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const int variant = mvi_attention_kernel_variant(Sq, Sk, D, dtype);
    if (variant != 0) {
        if (variant == 16)
            rc = dtype == MVI_DT_BF16 ? mvi::attn_flash8m16_launch<__hip_bfloat16>(q, k, v, out, B, H, Sq, Sk, scale, q_log2, st, q_ts, kv_ts, o_ts)
                                      : mvi::attn_flash8m16_launch<__half>(q, k, v, out, B, H, Sq, Sk, scale, q_log2, st, q_ts, kv_ts, o_ts);
        else if (variant == 8)
            rc = dtype == MVI_DT_BF16 ? mvi::attn_flash8_launch<__hip_bfloat16>(q, k, v, out, B, H, Sq, Sk, scale, q_log2, st, q_ts, kv_ts, o_ts)
                                      : mvi::attn_flash8_launch<__half>(q, k, v, out, B, H, Sq, Sk, scale, q_log2, st, q_ts, kv_ts, o_ts);
        else {
            const float sl2 = q_log2 ? 1.0f : scale * kLog2e;
            rc = dtype == MVI_DT_BF16 ? mvi::attn_flash_launch<__hip_bfloat16>(q, k, v, out, B, H, Sq, Sk, sl2, st, q_ts, kv_ts, o_ts)
                                      : mvi::attn_flash_launch<__half>(q, k, v, out, B, H, Sq, Sk, sl2, st, q_ts, kv_ts, o_ts);
        }
    } else {
        if (D != 16 && D != 32 && D != 64) return mvi::unet_fail(MVI_EINVAL, "attention: head dim must be 16, 32 or 64");
        switch (dtype) {
            case MVI_DT_F32: rc = mvi::attn_rowtile_launch<float>(q, k, v, out, B, H, Sq, Sk, D, scale, st, 0, q_ts, kv_ts, o_ts); break;
            case MVI_DT_BF16: rc = mvi::attn_rowtile_launch<__hip_bfloat16>(q, k, v, out, B, H, Sq, Sk, D, scale, st, 0, q_ts, kv_ts, o_ts); break;
            case MVI_DT_F16: rc = mvi::attn_rowtile_launch<__half>(q, k, v, out, B, H, Sq, Sk, D, scale, st, 0, q_ts, kv_ts, o_ts); break;
            default: return mvi::unet_fail(MVI_EINVAL, "attention: unknown dtype");
        }
    }
    return rc ? mvi::unet_fail(rc, "attention: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_attention_forward(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H,
                                     int32_t Sq, int32_t Sk, int32_t D, float scale, int32_t dtype, void* stream) {
    return attention_forward_impl(q, k, v, out, B, H, Sq, Sk, D, scale, dtype, 0, 0, 0, stream);
}

extern "C" int mvi_attention_forward_strided(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H,
                                             int32_t Sq, int32_t Sk, int32_t D, float scale, int32_t dtype,
                                             int64_t q_token_stride, int64_t kv_token_stride, int64_t out_token_stride,
                                             void* stream) {
    return attention_forward_impl(q, k, v, out, B, H, Sq, Sk, D, scale, dtype, q_token_stride, kv_token_stride,
                                  out_token_stride, stream);
}

extern "C" int mvi_attention_forward_strided_qlog2(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H,
                                                   int32_t Sq, int32_t Sk, int32_t D, int32_t dtype, int64_t q_token_stride,
                                                   int64_t kv_token_stride, int64_t out_token_stride, void* stream) {
    return attention_forward_impl(q, k, v, out, B, H, Sq, Sk, D, kLn2, dtype, q_token_stride, kv_token_stride, out_token_stride, stream, true);
}

namespace mvi {
bool attn_temporal16_ok(int T, int D, int dtype, int64_t hd, int64_t qkv_ts, int64_t o_ts, const void* q, const void* k, const void* v,
                        const void* out);
template <typename T>
int attn_temporal16_launch(const void* q, const void* k, const void* v, void* out, int Bo, int Tn, int S, int H, float scale, hipStream_t st,
                           int64_t qkv_ts, int64_t o_ts);
}  // namespace mvi

// 1 when the MFMA kernel of csrc/attn_temporal.hip serves this call, 0 for the fp32-math kernel of csrc/attn_rowtile.hip
extern "C" int mvi_attention_temporal_kernel_variant(int32_t T, int32_t H, int32_t D, int32_t dtype, int64_t qkv_token_stride,
                                                     int64_t out_token_stride) {
    static const bool off = getenv("MVI_ATTN_TEMPORAL_MFMA") && getenv("MVI_ATTN_TEMPORAL_MFMA")[0] == '0';     // same-box A/B runs
    return !off && mvi::attn_temporal16_ok(T, D, dtype, (int64_t)H * D, qkv_token_stride, out_token_stride, nullptr, nullptr, nullptr, nullptr);
}

static int attention_temporal_impl(const void* q, const void* k, const void* v, void* out, int32_t Bo, int32_t T, int32_t S,
                                   int32_t H, int32_t D, float scale, int32_t dtype, int64_t qkv_ts, int64_t o_ts, void* stream) {
    if (Bo < 0 || T <= 0 || S <= 0 || H <= 0 || D <= 0) return mvi::unet_fail(MVI_EINVAL, "temporal attention: bad shape");
    if (Bo == 0) return MVI_OK;
    if (!q || !k || !v || !out) return mvi::unet_fail(MVI_EINVAL, "temporal attention: NULL pointer");
    if (D != 16 && D != 32 && D != 64) return mvi::unet_fail(MVI_EINVAL, "temporal attention: head dim must be 16, 32 or 64");
    if ((int64_t)Bo * S > 0x7FFFFFFFll) return mvi::unet_fail(MVI_EINVAL, "temporal attention: too many problems");
    const int64_t hd = (int64_t)H * D;
    if (qkv_ts < 0 || o_ts < 0 || (qkv_ts && qkv_ts < hd) || (o_ts && o_ts < hd))
        return mvi::unet_fail(MVI_EINVAL, "temporal attention: token strides must be 0 or >= H*D elements");
    hipStream_t st = (hipStream_t)stream;
    const int B = Bo * S;
    int rc;
    if (mvi_attention_temporal_kernel_variant(T, H, D, dtype, qkv_ts, o_ts) && mvi::attn_temporal16_ok(T, D, dtype, hd, qkv_ts, o_ts, q, k, v, out)) {
        rc = dtype == MVI_DT_BF16 ? mvi::attn_temporal16_launch<__hip_bfloat16>(q, k, v, out, Bo, T, S, H, scale, st, qkv_ts, o_ts)
                                  : mvi::attn_temporal16_launch<__half>(q, k, v, out, Bo, T, S, H, scale, st, qkv_ts, o_ts);
        return rc ? mvi::unet_fail(rc, "temporal attention: kernel launch failed") : MVI_OK;
    }
    switch (dtype) {
        case MVI_DT_F32: rc = mvi::attn_rowtile_launch<float>(q, k, v, out, B, H, T, T, D, scale, st, S, qkv_ts, qkv_ts, o_ts); break;
        case MVI_DT_BF16: rc = mvi::attn_rowtile_launch<__hip_bfloat16>(q, k, v, out, B, H, T, T, D, scale, st, S, qkv_ts, qkv_ts, o_ts); break;
        case MVI_DT_F16: rc = mvi::attn_rowtile_launch<__half>(q, k, v, out, B, H, T, T, D, scale, st, S, qkv_ts, qkv_ts, o_ts); break;
        default: return mvi::unet_fail(MVI_EINVAL, "temporal attention: unknown dtype");
    }
    return rc ? mvi::unet_fail(rc, "temporal attention: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_attention_temporal(const void* q, const void* k, const void* v, void* out, int32_t Bo, int32_t T,
                                      int32_t S, int32_t H, int32_t D, float scale, int32_t dtype, void* stream) {
    return attention_temporal_impl(q, k, v, out, Bo, T, S, H, D, scale, dtype, 0, 0, stream);
}

extern "C" int mvi_attention_temporal_strided(const void* q, const void* k, const void* v, void* out, int32_t Bo, int32_t T,
                                              int32_t S, int32_t H, int32_t D, float scale, int32_t dtype,
                                              int64_t qkv_token_stride, int64_t out_token_stride, void* stream) {
    return attention_temporal_impl(q, k, v, out, Bo, T, S, H, D, scale, dtype, qkv_token_stride, out_token_stride, stream);
}

// q carries D^-1/2 log2(e): these kernels exponentiate with e, so ln 2 is the factor left to apply
extern "C" int mvi_attention_temporal_strided_qlog2(const void* q, const void* k, const void* v, void* out, int32_t Bo, int32_t T,
                                                    int32_t S, int32_t H, int32_t D, int32_t dtype, int64_t qkv_token_stride,
                                                    int64_t out_token_stride, void* stream) {
    return attention_temporal_impl(q, k, v, out, Bo, T, S, H, D, kLn2, dtype, qkv_token_stride, out_token_stride, stream);
}
