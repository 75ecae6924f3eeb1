// The 8-wave flash attention kernel of attn_flash8.hip on v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16 (round 6, VERDICT r5
// item 1): the same 256-query block, K/V LDS-DMA ring, barriers, fixed-exponent fast form and online-softmax safe form — only the
// matrix instruction and with it every lane layout differ. Why it exists: every linear_n320 form ran 9 - 11 % faster on the 16x16x32
// shape at equal cycles (the chip holds a higher clock under it: profiles/round5_n320_mfma16_ab.txt, MI355X_MICROARCH.md 'DVFS
// give-back' item 7), and attention was the one matrix kernel never built on it. Whether it pays HERE is a measurement
// (profiles/round6_attention_mfma16_ab.txt): this kernel is bound by vector issue, and an MFMA of either shape holds the issue port
// for 8 cycles — the 16x16x32 form issues twice as many of them per FLOP.
// Measured (profiles/round6_attention_mfma16_ab.txt, same box, alternating, q carrying the scale as the SVD modules run it): per 64-key
// tile and block 1638 shader cycles on 32x32x16, 1841 here, 1868 with the row sums as a fifth d tile (kOnes) — and an in-kernel clock of
// 1.83, 2.07 and 2.11 GHz: the chip is power-bound under this kernel and pays the cheaper instruction back as clock, 12 % more cycles
// become 2.7 - 3.0 % LESS time (S = 9216: 2.742 -> 2.668 ms, 0.443 -> 0.455 of 2.5 PF on that box; S = 2304: 0.402 -> 0.390 ms). On
// all-zero operands (2.39 GHz for all three) the order is the cycle order. So this kernel is the default since round 6
// (MVI_ATTN_MFMA16: 0 = attn_flash8.hip, 1 = row sums on the VALU, 2 = default); mvi_attention_kernel_variant reports 16.
// Replaces xformers.ops.memory_efficient_attention / SDPA (svd_inpaint1/sgm/modules/attention.py:427-439, :332-336).
//
// Layout: q/out [B, Sq, H, 64], k/v [B, Sk, H, 64] token-major with element strides between tokens, as in attn_flash8.hip.
// A wave owns 32 queries = two 16-query tiles (qt). Lane = (c = lane & 15, g = lane >> 4).
// Per wave and 32-key block (two 16-key tiles kt), v_mfma_f32_16x16x32, fp32 accumulate:
//   S'^T[key][query] (16 x 16) = K (16 keys x 32 d) Q'^T (32 d x 16 queries) - m : 2 kt x 2 qt x 2 d-steps = 8 MFMA
//        A = K rows from LDS: lane (c, g) holds K[16 kt + c][32 ks + 8 g .. + 7]  (one ds_read_b128, shared by both qt)
//        B = Q' in registers: lane (c, g) holds Q[16 qt + c][32 ks + 8 g .. + 7]
//        C/D: lane (c, g) holds S'^T[key 16 kt + 4 g + r][query 16 qt + c], r = 0..3; the first MFMA of a chain takes -m as C
//   P = exp2(S'): the query sits on the lane, everything lane-local; row sums reduced over g at the very end
//   O^T[d][query] (16 x 16) += V^T (16 d x 32 keys) P^T (32 keys x 16 queries): 4 d tiles x 2 qt = 8 MFMA
//        B = P from the S' accumulators: element j of lane (c, g) is k-index 8 g + j  <->  key (j < 4 ? 4 g + j : 16 + 4 g + j - 4)
//        A = V^T by two ds_read_b64_tr_b16 per d tile (keys 4 g .. 4 g + 3 of key tile 0, then of key tile 1): the same permutation
// LDS images (128-byte rows, 16-byte chunks XOR-swizzled on the DMA's SOURCE side): K chunk c of row r in slot c ^ ((r >> 1) & 7)
// (ds_read_b128 of 16 rows x one chunk: conflict-free); V chunk c of row r in slot c ^ (((r >> 1) & 3) << 1) (a transposed read
// touches 8 rows x 32 bytes per half wave: conflict-free).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {
namespace f8m {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define MVI_AS3 __attribute__((address_space(3)))

constexpr int kD = 64;            // head dim
constexpr int kKT = 64;           // keys per tile
constexpr int kRing = 4;
constexpr int kTileBytes = kKT * kD * 2;          // 8 KiB
constexpr int kLdsBytes = 2 * kRing * kTileBytes; // K ring | V ring = 64 KiB (+ 16 bytes: the block's "repeat safely" flag)
constexpr float kRescaleThreshold = 8.0f;         // log2 units
constexpr int kWaves = 8;
constexpr int kLoaders = 4;

template <typename T> struct Mma;
template <> struct Mma<__hip_bfloat16> {
    using frag = bf16x8;
    __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        bf16x2 r = __builtin_convertvector(f, bf16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
    __device__ static float lo(uint32_t w) { return __uint_as_float(w << 16); }
    __device__ static float hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
};
template <> struct Mma<__half> {
    using frag = f16x8;
    __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        f16x2 r = __builtin_convertvector(f, f16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
    __device__ static float lo(uint32_t w) { f16x2 h = *reinterpret_cast<f16x2*>(&w); return (float)h[0]; }
    __device__ static float hi(uint32_t w) { f16x2 h = *reinterpret_cast<f16x2*>(&w); return (float)h[1]; }
};

template <typename F> __device__ __forceinline__ F as_frag(u32x4 v) { return *reinterpret_cast<F*>(&v); }

// (as in attn_flash8.hip: the kernel counts its own vmcnt for the LDS-DMA pieces)
__device__ __forceinline__ void dma_piece(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// reduction over the four 16-lane groups of a wave (the keys of a query are spread over g)
__device__ __forceinline__ float group_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float group_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __builtin_fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// scores of one 32-key block: [key tile kt][query tile qt], each the C/D tile of one accumulation chain
struct Scores { f32x4 t[2][2]; };

// kOnes: the softmax row sums come out of the matrix pipe instead of the VALU — a fifth "d tile" whose V^T fragment is a row of ones
// (two more MFMAs per 32-key block, +1/8 of the matrix work) replaces the 32 v_add_f32 per wave and tile on the issue port that bounds
// this kernel; the sums are then those of the ROUNDED probabilities, i.e. of exactly the numerators the P V products use.
template <typename T, bool kExact, bool kOnes>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_flash8m16_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ out,
                           int H, int Sq, int Sk, float scale_log2e, int q_blocks, int total_blocks, int64_t q_rs,
                           int64_t kv_rs, int64_t o_rs) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int kQB = 32 * kWaves;             // query rows per block
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    MVI_AS3 char* const lds = (MVI_AS3 char*)smem;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);

    int bid = blockIdx.x;
    if ((total_blocks & 7) == 0) bid = (bid & 7) * (total_blocks >> 3) + (bid >> 3);    // consecutive q blocks of a head share an XCD's L2
    const int qb = bid % q_blocks;
    const int bh = bid / q_blocks;
    const int h = bh % H;
    const int64_t b = bh / H;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int qrow0 = qb * kQB + wave * 32 + c16;     // query of tile qt: qrow0 + 16 qt

    const float sc_mul = kExact ? scale_log2e : 1.0f;
    // ---- Q': B operand of S'^T = K Q'^T, element j of lane (c, g), query tile qt, d-step ks: Q[qrow0 + 16 qt][32 ks + 8 g + j]
    frag qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = qrow0 + 16 * qt;
        const T* qp = q + ((b * Sq + (qrow < Sq ? qrow : 0)) * q_rs + (int64_t)h * kD + 8 * g);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 raw = qrow < Sq ? *reinterpret_cast<const u32x4*>(qp + 32 * ks) : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (!kExact) raw[i] = M::pack2(M::lo(raw[i]) * scale_log2e, M::hi(raw[i]) * scale_log2e);
            qf[qt][ks] = as_frag<frag>(raw);
        }
    }

    // ---- LDS-DMA source addressing (attn_flash8.hip; only V's swizzle differs)
    const char* const kbase = reinterpret_cast<const char*>(k + (b * Sk * kv_rs + (int64_t)h * kD));
    const char* const vbase = reinterpret_cast<const char*>(v + (b * Sk * kv_rs + (int64_t)h * kD));
    constexpr int kMaxPieces = (16 + kLoaders - 1) / kLoaders;
    const int n_pieces = wave < kLoaders ? (16 - wave + kLoaders - 1) / kLoaders : 0;
    const uint32_t row_bytes = (uint32_t)(kv_rs * 2);
    const int pslot = lane & 7;
    int p_row[kMaxPieces], p_chunk[kMaxPieces];
    uint32_t p_voff[kMaxPieces], p_dst[kMaxPieces];
    bool p_is_v[kMaxPieces];
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
        const int pc = wave + i * kLoaders;
        p_is_v[i] = pc >= 8;
        p_row[i] = 8 * (pc & 7) + (lane >> 3);
        p_chunk[i] = p_is_v[i] ? pslot ^ (((p_row[i] >> 1) & 3) << 1) : pslot ^ ((p_row[i] >> 1) & 7);
        p_voff[i] = (uint32_t)p_row[i] * row_bytes + 16u * p_chunk[i];
        p_dst[i] = lds0 + (p_is_v[i] ? kRing * kTileBytes : 0) + 1024u * (pc & 7);
    }
    const int n_tiles = (Sk + kKT - 1) / kKT;
    const int n_full = Sk / kKT;
    auto issue_tile = [&](int tt) __attribute__((always_inline)) {
        const uint32_t ring_off = (uint32_t)((tt & (kRing - 1)) * kTileBytes);
        const bool full = tt < n_full;
        const int64_t off = (int64_t)tt * kKT * row_bytes;
#pragma unroll
        for (int i = 0; i < kMaxPieces; ++i) {
            if (i >= n_pieces) break;
            const char* const base = p_is_v[i] ? vbase : kbase;
            if (full) {
                dma_piece(base + off, p_voff[i], p_dst[i] + ring_off);
            } else {
                int r = tt * kKT + p_row[i];
                r = r < Sk ? r : Sk - 1;
                dma_piece(base, (uint32_t)r * row_bytes + 16u * p_chunk[i], p_dst[i] + ring_off);
            }
        }
    };
    auto wait_tiles_then_barrier = [&](auto tiles_c) __attribute__((always_inline)) {
        constexpr int kT = decltype(tiles_c)::value;
        static_assert(16 % kLoaders == 0, "every loader moves 16 / kLoaders pieces");
        if (n_pieces == 0) wait_vm_then_barrier<0>();
        else wait_vm_then_barrier<kMaxPieces * kT>();
    };

    // ---- LDS read addressing (per lane; ring slot / block / key tile enter as immediates)
    // K, A operand: row 32 kb + 16 kt + c, chunk 4 ks + g  ->  slot (4 ks + g) ^ ((c >> 1) & 7)   (16 kt and 32 kb leave (row >> 1) & 7 alone)
    uint32_t ka[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ka[ks] = (uint32_t)(c16 * 128 + (((4 * ks + g) ^ ((c16 >> 1) & 7)) << 4));
    // V^T, A operand of d tile dt: the 16-lane group g reads the 4-key x 16-d block (keys 4 g .. 4 g + 3 of a key tile, d0 = 16 dt);
    // lane 4 qq + p of the group supplies row qq, columns 4 p .. 4 p + 3 and receives column c. Row 4 g + qq holds chunk ch in slot
    // ch ^ (x << 1), x = ((4 g + qq) >> 1) & 3: byte 32 (dt ^ x) + 8 p of the row.
    uint32_t va[4];
    {
        const int qq = c16 >> 2, p = c16 & 3, x = (2 * (g & 1) + (qq >> 1)) & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) va[dt] = kRing * kTileBytes + (uint32_t)((4 * g + qq) * 128 + ((dt ^ x) << 5) + 8 * p);
    }

    f32x4 o[4][2], negm[2];                  // o[d tile][query tile]
    f32x4 lacc[2];                           // kOnes: row 0 of the "ones" tile = the row sums (lanes g == 0, register 0; zeros elsewhere)
    const uint32_t one2 = std::is_same<T, __half>::value ? 0x3C003C00u : 0x3F803F80u;
    const u32x4 ones = c16 == 0 ? u32x4{one2, one2, one2, one2} : u32x4{0, 0, 0, 0};     // A operand: V^T row 0 = 1 for every key
    Scores s0, s1;                           // scores of key block 0 / 1 of a tile: FIXED roles
    float l[2], rsum[2];

    // all of S' of one 32-key block (prologue only)
    auto qk_block = [&](int slot, int kb, Scores& sc) __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 kf = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ka[ks] + slot * kTileBytes + kb * 4096 + kt * 2048);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) sc.t[kt][qt] = M::mfma(as_frag<frag>(kf), qf[qt][ks], ks == 0 ? negm[qt] : sc.t[kt][qt]);
            }
    };
    // keys >= Sk of the block that starts at key kbase0 never win
    auto mask_block = [&](Scores& sc, int kbase0) __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kbase0 + 16 * kt + 4 * g + r >= Sk) { sc.t[kt][0][r] = -INFINITY; sc.t[kt][1][r] = -INFINITY; }
    };
    // Row max of a block whose scores already carry -m, for query tile qt
    auto block_max = [&](const Scores& sc, int qt) __attribute__((always_inline)) {
        const f32x4 a = sc.t[0][qt], bb = sc.t[1][qt];
        const float ra = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), __builtin_fmaxf(a[2], a[3]));
        const float rb = __builtin_fmaxf(__builtin_fmaxf(bb[0], bb[1]), __builtin_fmaxf(bb[2], bb[3]));
        return group_max(__builtin_fmaxf(ra, rb));
    };
    auto rescale = [&](Scores& sc, const bool (&grow)[2], const float (&rmax)[2], bool first) __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float delta = grow[qt] ? rmax[qt] : 0.f;
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta * sc_mul);
            l[qt] = (l[qt] + rsum[qt]) * alpha;
            rsum[qt] = 0.f;
            if (kOnes) lacc[qt] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
            sc.t[0][qt] -= delta;
            sc.t[1][qt] -= delta;
            negm[qt] -= delta;
        }
    };
    // K fragments of one key tile (both d-steps): two ds_read_b128
    struct KFrags { u32x4 k[2]; };
    auto load_k = [&](int slot, int kb, int kt, bool on) __attribute__((always_inline)) {
        KFrags f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            f.k[ks] = on ? *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ka[ks] + slot * kTileBytes + kb * 4096 + kt * 2048) : u32x4{0, 0, 0, 0};
        return f;
    };
    // V^T fragments of d tiles dt0, dt0 + 1 of block kb: four transposed reads into vs[dt0], vs[dt0 + 1]
    auto load_v = [&](u32x4 (&vs)[4], int slot, int kb, int dt0) __attribute__((always_inline)) {
#pragma unroll
        for (int dt = dt0; dt < dt0 + 2; ++dt) {
            const int off = slot * kTileBytes + kb * 4096;
            s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((MVI_AS3 s16x4*)(lds + va[dt] + off));
            s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((MVI_AS3 s16x4*)(lds + va[dt] + off + 2048));
            const u32x2 a = *reinterpret_cast<u32x2*>(&lo4), bb = *reinterpret_cast<u32x2*>(&hi4);
            vs[dt] = u32x4{a[0], a[1], bb[0], bb[1]};
        }
    };
    // exp / pack / row sum of query tile qt of a block: the P fragment (B operand of the P V MFMAs of that query tile)
    auto probs = [&](const Scores& sc, int qt) __attribute__((always_inline)) {
        u32x4 pr;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float x0 = kExact ? sc.t[kt][qt][2 * i] * sc_mul : sc.t[kt][qt][2 * i];
                const float x1 = kExact ? sc.t[kt][qt][2 * i + 1] * sc_mul : sc.t[kt][qt][2 * i + 1];
                const float p0 = __builtin_amdgcn_exp2f(x0);
                const float p1 = __builtin_amdgcn_exp2f(x1);
                if (!kOnes) rsum[qt] += p0 + p1;
                pr[2 * kt + i] = M::pack2(p0, p1);
            }
        if (!kOnes) asm volatile("" : "+v"(rsum[qt]));   // (as in attn_flash8.hip: the sum is complete here)
        return pr;
    };
    // The matrix work beside one quarter: the S' MFMAs of key tile kt of the OTHER block (into acc.t[kt][*]) and the four P V MFMAs
    // of query tile pqt with fragment pf and the V^T set vs. Dependent S' links sit two apart; two P V MFMAs trail the last link.
    auto matrix_part = [&](const KFrags& f, bool with_k, Scores& acc, int kt, u32x4 pf_raw, const u32x4 (&vs)[4], int pqt, bool with_pv) __attribute__((always_inline)) {
        const frag pf = as_frag<frag>(pf_raw);
        if (with_k) acc.t[kt][0] = M::mfma(as_frag<frag>(f.k[0]), qf[0][0], negm[0]);
        if (with_k) acc.t[kt][1] = M::mfma(as_frag<frag>(f.k[0]), qf[1][0], negm[1]);
        if (with_pv) o[0][pqt] = M::mfma(as_frag<frag>(vs[0]), pf, o[0][pqt]);
        if (with_k) acc.t[kt][0] = M::mfma(as_frag<frag>(f.k[1]), qf[0][1], acc.t[kt][0]);
        if (with_pv) o[1][pqt] = M::mfma(as_frag<frag>(vs[1]), pf, o[1][pqt]);
        if (with_k) acc.t[kt][1] = M::mfma(as_frag<frag>(f.k[1]), qf[1][1], acc.t[kt][1]);
        if (with_pv) o[2][pqt] = M::mfma(as_frag<frag>(vs[2]), pf, o[2][pqt]);
        if (with_pv) o[3][pqt] = M::mfma(as_frag<frag>(vs[3]), pf, o[3][pqt]);
        if (kOnes && with_pv) lacc[pqt] = M::mfma(as_frag<frag>(ones), pf, lacc[pqt]);
    };

    // The whole key loop for this block's 256 queries; kSafe as in attn_flash8.hip (false: the reference exponent is the row max of
    // the first 32 keys and never moves; true: online softmax, run only when the fast form's row sum left its range).
    auto run = [&](auto safe_c) __attribute__((always_inline)) {
        constexpr bool kSafe = decltype(safe_c)::value;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            negm[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            lacc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            l[qt] = 0.f;
            rsum[qt] = 0.f;
        }
        issue_tile(0);
        issue_tile(1);
        issue_tile(2);
        wait_tiles_then_barrier(std::integral_constant<int, 2>{});   // tile 0 landed everywhere
        qk_block(0, 0, s0);
        if (Sk < 32) mask_block(s0, 0);
        {
            const float rm[2] = {block_max(s0, 0), block_max(s0, 1)};
            const bool gr[2] = {true, true};
            rescale(s0, gr, rm, true);
        }
        wait_tiles_then_barrier(std::integral_constant<int, 1>{});   // tile 1
        u32x4 vb0[4], vb1[4];                                        // V^T sets of key block 0 / 1 of the current tile
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { vb0[dt] = u32x4{0, 0, 0, 0}; vb1[dt] = u32x4{0, 0, 0, 0}; }
        KFrags fq0 = load_k(0, 1, 0, true), fq1 = load_k(0, 1, 1, true);   // S' of tile 0, block 1
        load_v(vb0, 0, 0, 0);                                        // (d tiles 2, 3 follow in the first quarter)

        auto decide = [&](Scores& sc) __attribute__((always_inline)) {
            if (!kSafe) return;
            const float rm[2] = {block_max(sc, 0), block_max(sc, 1)};
            const bool gr[2] = {rm[0] * sc_mul > kRescaleThreshold, rm[1] * sc_mul > kRescaleThreshold};
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(gr[0] || gr[1]) != 0ull, 0)) rescale(sc, gr, rm, false);   // wave-uniform, rare
        };
        u32x4 pend = u32x4{0, 0, 0, 0};                              // fast form: the P fragment whose P V is still to be issued
        // One 64-key tile = four quarters; a quarter = one query tile of one 32-key block on the VALU (8 exp, 8 adds, 4 packs)
        // beside 8 MFMAs: the S' of one key tile of the other block and the P V of the PREVIOUS quarter's P (fast form) or of its
        // own (safe form).
        //            VALU                 S' MFMAs                      P V MFMAs (fast form)            LDS reads issued
        //   Q1   P(s0, qt 0)     s1 key tile 0 of tile t       P(prev s1, qt 1) x vb1(t - 1)     K(t+1, b0, kt0), vb0 d tiles 2, 3
        //   Q2   P(s0, qt 1)     s1 key tile 1                 P(s0, qt 0)     x vb0             K(t+1, b0, kt1), vb1 d tiles 0, 1
        //   Q3   P(s1, qt 0)     s0 key tile 0 of tile t + 1   P(s0, qt 1)     x vb0             K(t+1, b1, kt0), vb1 d tiles 2, 3
        //   Q4   P(s1, qt 1)     s0 key tile 1 of tile t + 1   P(s1, qt 0)     x vb1             K(t+1, b1, kt1), vb0(t+1) d tiles 0, 1
        constexpr bool kPipe = !kSafe;
        auto quarter = [&](const KFrags& f, bool with_k, Scores& acc, int kt, const Scores& sc, int qt, const u32x4 (&v_pipe)[4],
                           const u32x4 (&v_own)[4], bool pend_valid) __attribute__((always_inline)) {
            if (kPipe) {
                matrix_part(f, with_k, acc, kt, pend, v_pipe, qt ^ 1, pend_valid);
                pend = probs(sc, qt);
            } else {
                const u32x4 now = probs(sc, qt);
                matrix_part(f, with_k, acc, kt, now, v_own, qt, true);
            }
        };
        auto tile = [&](int t, auto slot_c, auto has_next_c) __attribute__((always_inline)) {
            const int slot = slot_c, next = (slot + 1) & (kRing - 1);
            const bool has_next = has_next_c;
            const int k0 = t * kKT;
            const bool ragged = !has_next && k0 + kKT > Sk;          // only the last tile can be ragged
            if (ragged && t > 0) mask_block(s0, k0);                 // (tile 0's first block was masked before it set m)
            decide(s0);
            KFrags f2 = load_k(next, 0, 0, has_next);
            load_v(vb0, slot, 0, 2);
            quarter(fq0, true, s1, 0, s0, 0, vb1, vb0, t > 0);       // (before tile 0 nothing is pending)
            __builtin_amdgcn_sched_barrier(0);
            KFrags f3 = load_k(next, 0, 1, has_next);
            load_v(vb1, slot, 1, 0);
            quarter(fq1, true, s1, 1, s0, 1, vb0, vb0, true);
            __builtin_amdgcn_sched_barrier(0);
            if (ragged) mask_block(s1, k0 + 32);
            decide(s1);
            if (has_next) fq0 = load_k(next, 1, 0, true);
            load_v(vb1, slot, 1, 2);
            quarter(f2, has_next, s0, 0, s1, 0, vb0, vb1, true);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) fq1 = load_k(next, 1, 1, true);
            if (has_next) load_v(vb0, next, 0, 0);
            quarter(f3, has_next, s0, 1, s1, 1, vb1, vb1, true);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                // slot (t + 3) % 4 held tile t - 1: nobody reads it after the barrier that opened this tile
                issue_tile(t + 3);
                wait_tiles_then_barrier(std::integral_constant<int, 1>{});   // own pieces of tile t + 2 (and everything older) landed
            }
        };
        using std::integral_constant;
        using std::true_type;
        int t = 0;
        for (; t + 4 < n_tiles; t += 4) {
            tile(t, integral_constant<int, 0>{}, true_type{});
            tile(t + 1, integral_constant<int, 1>{}, true_type{});
            tile(t + 2, integral_constant<int, 2>{}, true_type{});
            tile(t + 3, integral_constant<int, 3>{}, true_type{});
        }
        for (; t < n_tiles; ++t) tile(t, t & (kRing - 1), t + 1 < n_tiles);
        if (kPipe) {                                                 // the last quarter's P V: P(s1, qt 1) x vb1
            KFrags none;
            none.k[0] = none.k[1] = u32x4{0, 0, 0, 0};
            matrix_part(none, false, s0, 0, pend, vb1, 1, true);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // trailing (unused) pieces must land before the ring is reused / released
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) l[qt] = group_sum(kOnes ? lacc[qt][0] : l[qt] + rsum[qt]);   // the four lane groups hold disjoint keys (kOnes: the sum sits in group 0, zeros elsewhere)
    };

    MVI_AS3 uint32_t* const redo_flag = (MVI_AS3 uint32_t*)(lds + kLdsBytes);
    if (tid == 0) *redo_flag = 0u;                               // ordered before any read by the barriers of run()
    run(std::false_type{});
    const float l_limit = std::is_same<T, __half>::value ? 0x1p15f : 0x1p100f;
    // Compared as BIT PATTERNS: with the row sums from the matrix pipe (kOnes) a probability that overflowed the type (f16: > 65504)
    // gives 0 x inf = NaN in the rows of the ones tile that hold zeros, i.e. a NaN sum — and this file is compiled with
    // -fno-honor-nans, under which `!(l <= limit)` may be folded to `l > limit` (false for NaN). Sums are >= 0: |bits| orders them,
    // and every inf / NaN pattern is above every finite limit.
    const uint32_t lim = __float_as_uint(l_limit);
    const bool out_of_range = (__float_as_uint(l[0]) & 0x7FFFFFFFu) > lim || (__float_as_uint(l[1]) & 0x7FFFFFFFu) > lim;
    if (__builtin_amdgcn_ballot_w64(out_of_range) != 0ull && lane == 0) *redo_flag = 1u;
    __syncthreads();
    if (*redo_flag != 0u) {
        __syncthreads();
        run(std::true_type{});
    }

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = qrow0 + 16 * qt;
        if (qrow < Sq) {
            const float inv = 1.0f / l[qt];
            T* op = out + ((b * Sq + qrow) * o_rs + (int64_t)h * kD + 4 * g);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 w = {M::pack2(o[dt][qt][0] * inv, o[dt][qt][1] * inv), M::pack2(o[dt][qt][2] * inv, o[dt][qt][3] * inv)};
                *reinterpret_cast<u32x2*>(op + 16 * dt) = w;
            }
        }
    }
}

}  // namespace f8m

template <typename T>
int attn_flash8m16_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk,
                          float scale, bool q_log2, hipStream_t st, int64_t q_rs, int64_t kv_rs, int64_t o_rs) {
    using namespace f8m;
    const int64_t hd = (int64_t)H * kD;
    if (q_rs == 0) q_rs = hd;
    if (kv_rs == 0) kv_rs = hd;
    if (o_rs == 0) o_rs = hd;
    constexpr int kQB = 32 * kWaves;
    const int q_blocks = (Sq + kQB - 1) / kQB;
    const int64_t total = (int64_t)B * H * q_blocks;
    if (total > 0x7FFFFFFFll) return MVI_EINVAL;
    if ((int64_t)Sk * kv_rs * 2 > 0xFFFFFFFFll) return MVI_EINVAL;       // 32-bit byte offsets inside one batch entry
    static unsigned long long attr_set = 0ull;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MVI_EHIP;
    static const int fold_env = getenv("MVI_ATTN_FOLD_SCALE") ? atoi(getenv("MVI_ATTN_FOLD_SCALE")) : -1;
    const bool fold = q_log2 || (fold_env >= 0 ? fold_env != 0 : std::is_same<T, __half>::value);     // (attn_flash8.hip)
    static const int mode = getenv("MVI_ATTN_MFMA16") ? atoi(getenv("MVI_ATTN_MFMA16")) : 2;          // 1: row sums on the VALU (A/B), otherwise on the matrix pipe
    const bool ones = mode != 1;
    if (!((attr_set >> dev) & 1ull)) {
        const void* all[4] = {reinterpret_cast<const void*>(&attn_flash8m16_kernel<T, true, false>), reinterpret_cast<const void*>(&attn_flash8m16_kernel<T, false, false>),
                              reinterpret_cast<const void*>(&attn_flash8m16_kernel<T, true, true>), reinterpret_cast<const void*>(&attn_flash8m16_kernel<T, false, true>)};
        for (const void* f : all)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes + 16) != hipSuccess) return MVI_EHIP;
        attr_set |= 1ull << dev;
    }
    auto kern = ones ? (fold ? &attn_flash8m16_kernel<T, false, true> : &attn_flash8m16_kernel<T, true, true>)
                     : (fold ? &attn_flash8m16_kernel<T, false, false> : &attn_flash8m16_kernel<T, true, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64 * kWaves), kLdsBytes + 16, st, (const T*)q, (const T*)k, (const T*)v,
                       (T*)out, H, Sq, Sk, q_log2 ? 1.0f : scale * 1.4426950408889634f, q_blocks, (int)total, q_rs, kv_rs, o_rs);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
template int attn_flash8m16_launch<__hip_bfloat16>(const void*, const void*, const void*, void*, int, int, int, int, float, bool, hipStream_t, int64_t, int64_t, int64_t);
template int attn_flash8m16_launch<__half>(const void*, const void*, const void*, void*, int, int, int, int, float, bool, hipStream_t, int64_t, int64_t, int64_t);

}  // namespace mvi
