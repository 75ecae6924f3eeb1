// bf16/f16 MFMA flash attention (forward), head dim 64, 8-wave workgroups — the kernel behind the large spatial
// self-attentions of the SVD denoise step: (B*H, S) = (140, 9216) and (280, 2304) at 14 x 576x1024, 99.7 % of the
// attention FLOPs. Replaces xformers.ops.memory_efficient_attention / SDPA
// (svd_inpaint1/sgm/modules/attention.py:427-439, :332-336). attn_flash.hip keeps the 4-wave kernel for short sequences.
//
// Why a second kernel: at D = 64 the 4-wave kernel is bound by the SIMD's ISSUE port, not by the matrix pipe
// (profiles/r01u_pmc_attention.txt: 13.8 VALU per MFMA, pipe 39.7 % busy). This one removes issue slots:
//   * the running reference exponent m never touches the VALU: -m enters the scores as the C operand of the first MFMA of every
//     QK^T chain (a 16-register block holding -m). The softmax SCALE has two forms (template kExact, below): applied to the fp32
//     scores (one v_mul per score; the form a caller's own q gets in bf16), or absent from the loop because Q carries
//     scale * log2(e) — either rounded into Q in the prologue (f16's default: a second rounding of q) or, since round 5, already
//     inside q from its projection's weights (mvi_attention_forward_strided_qlog2: no second rounding, both types; what the
//     self-attentions of the SVD modules run) — so P = exp2(S') directly;
//   * K and V tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, 2 per wave and
//     tile) instead of through VGPRs (8 loads + 12 LDS stores + 16 pack ops per wave and tile in the 4-wave kernel);
//   * V stays row-major in LDS and is read transposed by ds_read_b64_tr_b16 (no hand transposition);
//   * the loop is unrolled over a 4-deep ring with two score accumulators swapping roles, so neither ring offsets
//     nor the accumulator hand-over cost instructions (16 v_mov_b64 per tile in the 4-wave kernel);
//   * 256 query rows share one K/V tile (8 waves x 32 rows): half the LDS fill traffic per row.
//
// Layout: q/out [B, Sq, H, 64], k/v [B, Sk, H, 64] token-major with element strides q_rs / kv_rs / o_rs between tokens
// (H*64 when contiguous, 3*H*64 inside a packed projection).
// Per wave and 64-key tile (v_mfma_f32_32x32x16, fp32 accumulate):
//   S'^T[key][query] = K Q'^T - m : 2 key blocks x 4 d-steps = 8 MFMA (A = K rows from LDS, B = Q' in registers)
//   P = exp2(S'), row sums, bf16/f16 pack: the query sits on the lane, so everything is lane-local but one half swap
//   O^T[d][query] += V^T P^T      : 2 d blocks x 4 key-steps = 8 MFMA (A = V^T by transposed LDS reads, B = P from the
//                                   S' accumulators: MFMA C/D layout == next B layout up to a fixed key permutation
//                                   that the V^T reads follow)
// LDS images (128-byte rows, 16-byte chunk index XOR-swizzled; the DMA's destination is lane-linear, so the swizzle is
// applied to the SOURCE address): K chunk c of row r sits in slot c ^ ((r >> 1) & 7) (ds_read_b128 conflict-free);
// V chunk c of row r in slot c ^ (((r >> 1) & 1) << 2) (ds_read_b64_tr_b16 conflict-free).
// Ring: 4 slots each for K and V; iteration t (between two barriers) issues K_{t+3}, V_{t+3}, computes
// S' of the next 32 keys (matrix pipe) under the softmax of the current 32 (VALU) and their P V; before the closing barrier
// every wave waits for its own pieces of K_{t+2}, V_{t+2} with a COUNTED vmcnt (the two pieces of tile t+3 stay in flight
// across the barrier).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {
namespace f8 {

typedef __attribute__((ext_vector_type(16))) float f32x16;
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
constexpr float kRescaleThreshold = 8.0f;         // log2 units: O, l rescaled only when the row max grows by > 2^8

template <typename T> struct Mma;
template <> struct Mma<__hip_bfloat16> {
    using frag = bf16x8;
    __device__ static f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {     // one v_cvt_pk_bf16_f32 (RNE)
        f32x2 f = {lo, hi};
        bf16x2 r = __builtin_convertvector(f, bf16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
    __device__ static float lo(uint32_t w) { return __uint_as_float(w << 16); }
    __device__ static float hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
};
template <> struct Mma<__half> {
    using frag = f16x8;
    __device__ static f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        f16x2 r = __builtin_convertvector(f, f16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
    __device__ static float lo(uint32_t w) { f16x2 h = *reinterpret_cast<f16x2*>(&w); return (float)h[0]; }
    __device__ static float hi(uint32_t w) { f16x2 h = *reinterpret_cast<f16x2*>(&w); return (float)h[1]; }
};

template <typename F> __device__ __forceinline__ F as_frag(u32x4 v) { return *reinterpret_cast<F*>(&v); }

// One LDS-DMA piece: every lane moves 16 bytes from sbase + voff to LDS address (m0 + 16 * lane). Invisible to the
// compiler's wait-count bookkeeping on purpose: the kernel counts its own vmcnt (a builtin DMA makes hipcc put
// s_waitcnt vmcnt(0) in front of every later LDS read, which serialises the ring).
__device__ __forceinline__ void dma_piece(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// kWaves = 8: one 512-thread block per CU (2 waves per SIMD, 256 query rows share a K/V tile). kLoaders = 4: waves 0 .. 3
// move the LDS-DMA pieces, at the END of a tile (see issue_tile). The timing ablations, in-kernel stamps and block timeline
// that led here (other kWaves / kLoaders, static wave priority, DMA pieces spread over the quarters — all measured and not kept) are in
// profiles/HISTORY.md rounds 2 - 5 and in this file's git history; the diagnostic build that exists today is generated from this file
// (tools/attn_dev/build_stamped.sh), not kept beside it.
// kExact: the softmax scale is applied to the fp32 scores (one v_mul per score) instead of being rounded into Q. Folding
// scale * log2(e) into Q saves those 32 multiplies per wave and tile but rounds Q a second time to bf16: an error of
// |logit| * 2^-9 in the exponent, i.e. a few per cent on P where two keys with logits of ~60 compete (2.7e-2 of the
// output scale on the adversarial rows of tests/test_unet_ops_gpu.py, against 5e-3 with the exact form).
constexpr int kWaves = 8;
constexpr int kLoaders = 4;

template <typename T, bool kExact>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_flash8_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ out,
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
    const int qcol = lane & 31, hh = lane >> 5;
    const int qrow = qb * kQB + wave * 32 + qcol;

    const float sc_mul = kExact ? scale_log2e : 1.0f;            // what a score is multiplied by on its way into exp2
    // ---- Q' = Q (exact form) or round(Q * scale * log2 e): B operand of S^T = K Q^T, element j of lane (qcol, hh), d-step s: Q[qrow][16 s + 8 hh + j]
    frag qf[4];
    {
        const T* qp = q + ((b * Sq + (qrow < Sq ? qrow : 0)) * q_rs + (int64_t)h * kD + 8 * hh);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 raw = qrow < Sq ? *reinterpret_cast<const u32x4*>(qp + 16 * s) : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (!kExact) raw[i] = M::pack2(M::lo(raw[i]) * scale_log2e, M::hi(raw[i]) * scale_log2e);
            qf[s] = as_frag<frag>(raw);
        }
    }

    // ---- LDS-DMA source addressing. A tile is 16 pieces of 1 KiB: pieces 0..7 = K rows 8 p .. 8 p + 7, pieces 8..15 = the
    // same for V. Wave w moves pieces w, w + kWaves, ... (< 16); lane i of piece p fills LDS slot (row 8 (p & 7) + (i >> 3),
    // 16-byte slot i & 7) with the chunk the image's swizzle assigns to that slot.
    const char* const kbase = reinterpret_cast<const char*>(k + (b * Sk * kv_rs + (int64_t)h * kD));
    const char* const vbase = reinterpret_cast<const char*>(v + (b * Sk * kv_rs + (int64_t)h * kD));
    // kLoaders < kWaves: only waves 0 .. kLoaders - 1 move pieces, and they do it at the END of a tile. Issuing an LDS-DMA
    // piece costs its wave ~150 cycles (in-kernel stamps: ~300 of a tile's ~1730 cycles per wave sat between the barrier and
    // the first quarter), and the first-dispatched half of the workgroup wins the SIMD's arbitration and waits ~500 cycles
    // per tile at the barrier for the other half: that wait is where the loaders issue.
    constexpr int kMaxPieces = (16 + kLoaders - 1) / kLoaders;
    const int n_pieces = wave < kLoaders ? (16 - wave + kLoaders - 1) / kLoaders : 0;   // pieces this wave moves per tile (wave-uniform)
    const uint32_t row_bytes = (uint32_t)(kv_rs * 2);
    const int pslot = lane & 7;
    int p_row[kMaxPieces], p_chunk[kMaxPieces];
    uint32_t p_voff[kMaxPieces], p_dst[kMaxPieces];
    bool p_is_v[kMaxPieces];
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
        const int pc = wave + i * kLoaders;                          // wave-uniform (meaningless for a wave that loads nothing)
        p_is_v[i] = pc >= 8;
        p_row[i] = 8 * (pc & 7) + (lane >> 3);
        p_chunk[i] = p_is_v[i] ? pslot ^ (((p_row[i] >> 1) & 1) << 2) : pslot ^ ((p_row[i] >> 1) & 7);
        p_voff[i] = (uint32_t)p_row[i] * row_bytes + 16u * p_chunk[i];
        p_dst[i] = lds0 + (p_is_v[i] ? kRing * kTileBytes : 0) + 1024u * (pc & 7);
    }
    const int n_tiles = (Sk + kKT - 1) / kKT;
    const int n_full = Sk / kKT;                 // tiles whose 64 rows all exist
    // tiles past the end are still "loaded" (rows clamped to Sk - 1) so that every iteration issues the same number of
    // pieces and the counted vmcnt stays valid; their ring slots are never read
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
    // counted waits: "at most `tiles_in_flight` tiles' worth of this wave's own pieces still outstanding", then the barrier
    auto wait_tiles_then_barrier = [&](auto tiles_c) __attribute__((always_inline)) {
        constexpr int kT = decltype(tiles_c)::value;
        static_assert(16 % kLoaders == 0, "every loader moves 16 / kLoaders pieces");
        if (n_pieces == 0) wait_vm_then_barrier<0>();
        else wait_vm_then_barrier<kMaxPieces * kT>();
    };

    // ---- LDS read addressing (per lane, ring slot / key block / k-step enter as immediates)
    // K, A operand of S^T: row 32 kb + qcol, chunk 2 s + hh  ->  slot (2 s + hh) ^ ((qcol >> 1) & 7)
    uint32_t ka[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = (uint32_t)(qcol * 128 + (((2 * s + hh) ^ ((qcol >> 1) & 7)) << 4));
    // V^T, A operand of O^T: 16-lane group g = lane >> 4 reads the 4-key x 16-d block (keys K0 + 4 hh + 0..3, d0 = 32 db +
    // 16 (g & 1)); lane 4 qq + p of the group supplies row qq, columns 4 p .. 4 p + 3 and receives column (lane & 15)
    uint32_t va[2];
    {
        const int i16 = lane & 15, qq = i16 >> 2, p = i16 & 3, g1 = (lane >> 4) & 1, sel = (qq >> 1) & 1;
        const uint32_t base = (uint32_t)((4 * hh + qq) * 128 + ((2 * g1 + (p >> 1)) << 4) + 8 * (p & 1));
        va[0] = kRing * kTileBytes + base + 64u * sel;          // d block 0: chunks 0..3 ^ swizzle
        va[1] = kRing * kTileBytes + base + 64u * (1 - sel);    // d block 1: chunks 4..7 ^ swizzle
    }

    f32x16 o[2], negm, s0, s1;               // s0 / s1: scores of key block 0 / 1 of a tile: FIXED roles, nothing is ever handed over
    float l, rsum;                           // l: row sum up to the last rescale; rsum: what was added since

    // S' of one 32-key block: rows 32 kb .. 32 kb + 31 of K ring slot `slot`, d-steps s_lo .. s_lo + 1 (two of the
    // chain's four MFMAs; the first takes -m as its C operand)
    auto qk2 = [&](int slot, int kb, int s_lo, f32x16& sc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = s_lo; s < s_lo + 2; ++s) {
            const u32x4 kf = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ka[s] + slot * kTileBytes + kb * 4096);
            sc = M::mfma(as_frag<frag>(kf), qf[s], s == 0 ? negm : sc);
        }
    };
    // Row max of a 32-key block whose scores already carry -m (both lane halves of a query hold disjoint keys)
    auto block_max = [&](const f32x16& sc) __attribute__((always_inline)) {
        float ra = __builtin_fmaxf(sc[0], sc[1]), rb = __builtin_fmaxf(sc[2], sc[3]);
#pragma unroll
        for (int r = 4; r < 16; r += 4) {                        // two independent v_max3 chains
            ra = __builtin_fmaxf(__builtin_fmaxf(ra, sc[r]), sc[r + 1]);
            rb = __builtin_fmaxf(__builtin_fmaxf(rb, sc[r + 2]), sc[r + 3]);
        }
        const float rmax = __builtin_fmaxf(ra, rb);
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
        return __builtin_fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    // Moves the reference exponent of the lanes in `grow` up by their block's excess over it: O, l, the block's scores
    // and the -m operand all follow (first: O = l = 0, nothing to scale — and 0 * 2^big would be NaN)
    auto rescale = [&](f32x16& sc, bool grow, float rmax, bool first) __attribute__((always_inline)) {
        const float delta = grow ? rmax : 0.f;               // in score units (the scores carry -m in the same units)
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta * sc_mul);
        l = (l + rsum) * alpha;
        rsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o[0][i] *= alpha; o[1][i] *= alpha;
            sc[i] -= delta;
            negm[i] -= delta;
        }
    };
    // LDS fragments of one 16-key quarter: two K row-fragments for the OTHER block's S' MFMAs and the four transposed
    // V reads of this quarter's P V MFMAs. Loaded one quarter ahead of their use, so no MFMA waits on an LDS read.
    // The S' chain of a block is split 3 + 1 over the two quarters that run beside it: its last MFMA then has the second
    // quarter's two P V MFMAs behind it before the next step's exp reads the scores (an exp issued right behind the chain
    // waits out the MFMA's latency, and so does the whole in-order wave).
    struct Frags { u32x4 k[3]; u32x2 v[2][2]; };
    auto load_frags = [&](int kslot, int kb, int s_lo, bool with_k, int vslot, int vkeys) __attribute__((always_inline)) {
        Frags f;
        const int nk = s_lo == 0 ? 3 : 1;    // d-steps 0, 1, 2 in the first quarter, 3 in the second
#pragma unroll
        for (int i = 0; i < 3; ++i)
            f.k[i] = (with_k && i < nk) ? *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ka[s_lo + i] + kslot * kTileBytes + kb * 4096) : u32x4{0, 0, 0, 0};
        // element j of the P fragment is key vkeys + 8 (j >> 2) + 4 hh + (j & 3): two transposed 4-key reads per d block
        const int koff = vslot * kTileBytes + vkeys * 128;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((MVI_AS3 s16x4*)(lds + va[db] + koff));
            s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((MVI_AS3 s16x4*)(lds + va[db] + koff + 8 * 128));
            f.v[db][0] = *reinterpret_cast<u32x2*>(&lo4);
            f.v[db][1] = *reinterpret_cast<u32x2*>(&hi4);
        }
        return f;
    };
    // exp / pack / row sum of registers 8 s2 .. 8 s2 + 7 of `sc`: the P fragment of 16 keys
    auto probs = [&](int s2, const f32x16& sc) __attribute__((always_inline)) {
        u32x4 pr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x0 = kExact ? sc[8 * s2 + 2 * i] * sc_mul : sc[8 * s2 + 2 * i];
            const float x1 = kExact ? sc[8 * s2 + 2 * i + 1] * sc_mul : sc[8 * s2 + 2 * i + 1];
            const float p0 = __builtin_amdgcn_exp2f(x0);
            const float p1 = __builtin_amdgcn_exp2f(x1);
            rsum += p0 + p1;
            pr[i] = M::pack2(p0, p1);
        }
        asm volatile("" : "+v"(rsum));       // the sum is complete HERE: without this, the fast form (which reads rsum only
                                             // after the loop) sinks four tiles' adds — and 128 live P values — to the loop's end
        return pr;
    };
    // What a quarter leaves for the next one to multiply into O: its P fragment and its four V^T fragments
    struct Pending { u32x4 p; u32x2 v[2][2]; };
    // The S' MFMAs that run beside a quarter (one accumulator chain per 32-key block: d-steps 0, 1, 2 beside the block's
    // first quarter, d-step 3 beside its second) with the two P V MFMAs of `pd` between them
    auto matrix_part = [&](const Frags& f, bool with_k, int s_lo, f32x16& acc, const Pending& pd, bool with_pv) __attribute__((always_inline)) {
        const u32x4 av0 = {pd.v[0][0][0], pd.v[0][0][1], pd.v[0][1][0], pd.v[0][1][1]};
        const u32x4 av1 = {pd.v[1][0][0], pd.v[1][0][1], pd.v[1][1][0], pd.v[1][1][1]};
        const frag pf = as_frag<frag>(pd.p);
        const bool qk = with_k;
        // The S' MFMAs of a block are ONE accumulator chain (3 + 1 over its two quarters) with the P V MFMAs between the
        // links; the scheduler spreads the quarter's exp / add / pack between them. Measured and not kept: 2 + 2 links per
        // quarter ordered S', PV, PV, S' behind scheduling fences (two independent MFMAs between dependent ones): the
        // fences also keep the VALU work out of the gaps, a lone wave went from 857 to 1147 cycles per tile.
        if (s_lo == 0) {
            if (qk) acc = M::mfma(as_frag<frag>(f.k[0]), qf[0], negm);
            if (with_pv) o[0] = M::mfma(as_frag<frag>(av0), pf, o[0]);
            if (qk) acc = M::mfma(as_frag<frag>(f.k[1]), qf[1], acc);
            if (with_pv) o[1] = M::mfma(as_frag<frag>(av1), pf, o[1]);
            if (qk) acc = M::mfma(as_frag<frag>(f.k[2]), qf[2], acc);
        } else {
            if (qk) acc = M::mfma(as_frag<frag>(f.k[0]), qf[3], acc);
            if (with_pv) o[0] = M::mfma(as_frag<frag>(av0), pf, o[0]);
            if (with_pv) o[1] = M::mfma(as_frag<frag>(av1), pf, o[1]);
        }
    };
    // One quarter of the SAFE form: P of these 16 keys, then the S' MFMAs beside it and its own two P V MFMAs (the rescale
    // of the next block's decision must find nothing pending).
    // One quarter of the FAST form (software-pipelined): the MFMAs of this region are the S' MFMAs and the P V MFMAs of
    // the PREVIOUS quarter (`pend`), the VALU work is the P of THIS quarter — independent of each other, so the scheduler
    // can put a handful of exp / add / pack behind every MFMA and an in-order wave never waits for its own pack before a
    // matrix instruction. (A lone wave of the un-pipelined form needed 1495 cycles per 64-key tile against ~570 of issue
    // and 512 of matrix-pipe work: tools/attn_dev/clk.sh.)
    auto quarter = [&](auto pipelined_c, const Frags& f, bool with_k, int s_lo, f32x16& acc, int s2, const f32x16& sc,
                       Pending& pend, bool pend_valid) __attribute__((always_inline)) {
        constexpr bool kPipe = decltype(pipelined_c)::value;
        if (kPipe) {
            matrix_part(f, with_k, s_lo, acc, pend, pend_valid);
            Pending nx;
            nx.p = probs(s2, sc);
            nx.v[0][0] = f.v[0][0]; nx.v[0][1] = f.v[0][1]; nx.v[1][0] = f.v[1][0]; nx.v[1][1] = f.v[1][1];
            pend = nx;
        } else {
            Pending now;
            now.p = probs(s2, sc);
            now.v[0][0] = f.v[0][0]; now.v[0][1] = f.v[0][1]; now.v[1][0] = f.v[1][0]; now.v[1][1] = f.v[1][1];
            matrix_part(f, with_k, s_lo, acc, now, true);
        }
    };

    // The whole key loop for this block's 256 queries. Two forms of the SAME arithmetic:
    //   kSafe = false (tried first): the reference exponent m is the row max of the FIRST 32 keys and never moves. Nothing
    //     about softmax needs the true maximum: P = 2^(s - m) / sum is exact for any m as long as nothing over- or
    //     underflows, fp32 / bf16 keep their relative precision at any magnitude, and keys far below m are as negligible
    //     against the first block's own maximum (P = 1) as against the true one. Dropping the per-block max takes 14 v_max3
    //     + a half swap + a compare-and-branch per tile off the issue port that bounds this kernel.
    //   kSafe = true: the usual online softmax (max per 32-key block, rescale when it grows by more than 2^8). Run only if
    //     the fast form's row sum left [0, 2^100] ([0, 2^15] in f16, where P must fit the type; or became NaN) for any query of the block — i.e. some score exceeded the
    //     first block's maximum by ~100 / log2(e) = 69 — everything is then recomputed from scratch with this form.
    auto run = [&](auto safe_c) __attribute__((always_inline)) {
        constexpr bool kSafe = decltype(safe_c)::value;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; negm[i] = 0.f; }
        l = 0.f;
        rsum = 0.f;
        // prologue: tiles 0, 1, 2 in flight; scores of block 0 of tile 0, whose row max becomes m
        issue_tile(0);
        issue_tile(1);
        issue_tile(2);
        wait_tiles_then_barrier(std::integral_constant<int, 2>{});   // tile 0 landed everywhere
        qk2(0, 0, 0, s0);
        qk2(0, 0, 2, s0);
        if (Sk < 32) {                           // (only when the whole problem is one ragged block)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (((r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) s0[r] = -INFINITY;
        }
        rescale(s0, true, block_max(s0), true);
        wait_tiles_then_barrier(std::integral_constant<int, 1>{});   // tile 1
        Frags fq0 = load_frags(0, 1, 0, true, 0, 0), fq1 = load_frags(0, 1, 3, true, 0, 16);   // first two quarters of tile 0

        // One 64-key tile = two steps of 32 keys, software-pipelined at HALF-tile granularity so the two score
        // accumulators never change roles (a whole-tile pipeline swaps them every tile and the register allocator pays
        // for that with accumulator copies):
        //   step 1: matrix pipe  s1 = S'(t, keys 32..63)     | VALU  softmax(s0 = S'(t, keys 0..31)), then P V of those keys
        //   step 2: matrix pipe  s0 = S'(t + 1, keys 0..31)  | VALU  softmax(s1),                     then P V of those keys
        // In the main loop the ring slot (t % kRing) and has_next are compile-time constants: ring offsets are instruction
        // immediates, and the MFMAs of the other block sit in the same scheduling region as this block's exp / pack. The
        // scheduling fences keep the 16-key quarters apart (left alone, the scheduler hoists every LDS read of a tile and
        // the allocator runs out of registers); inside a quarter the NEXT quarter's fragments are requested first.
        auto decide = [&](f32x16& sc) __attribute__((always_inline)) {
            if (!kSafe) return;
            const float rmax = block_max(sc);
            const bool grow = rmax * sc_mul > kRescaleThreshold;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(grow) != 0ull, 0)) rescale(sc, grow, rmax, false);   // wave-uniform, rare
        };
        // Fragments are requested TWO quarters before their use (measured: an MFMA that waits on its K fragment costs 19 %
        // of the kernel), also across the tile boundary: while tile t runs, tiles t and t + 1 are complete in LDS (the
        // closing wait of tile t - 1 covered tile t + 1), so its last two quarters request the first two of tile t + 1.
        Pending pend;                                            // fast form: the quarter whose P V is still to be issued
        pend.p = u32x4{0, 0, 0, 0};
        pend.v[0][0] = pend.v[0][1] = pend.v[1][0] = pend.v[1][1] = u32x2{0, 0};
        constexpr std::integral_constant<bool, !kSafe> pipe_c{};
        auto tile = [&](int t, auto slot_c, auto has_next_c) __attribute__((always_inline)) {
            const int slot = slot_c, next = (slot + 1) & (kRing - 1);
            const bool has_next = has_next_c;
            const int k0 = t * kKT;
            const bool ragged = !has_next && k0 + kKT > Sk;      // only the last tile can be ragged: keys >= Sk never win
            if (ragged && t > 0) {                               // (tile 0's first block was masked before it set m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((k0 + (r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) s0[r] = -INFINITY;
            }
            decide(s0);
            Frags f2 = load_frags(next, 0, 0, has_next, slot, 32);
            quarter(pipe_c, fq0, true, 0, s1, 0, s0, pend, t > 0);          // (before tile 0 nothing is pending)
            __builtin_amdgcn_sched_barrier(0);
            Frags f3 = load_frags(next, 0, 3, has_next, slot, 48);
            quarter(pipe_c, fq1, true, 3, s1, 1, s0, pend, true);
            __builtin_amdgcn_sched_barrier(0);
            if (ragged) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((k0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) s1[r] = -INFINITY;
            }
            decide(s1);
            if (has_next) fq0 = load_frags(next, 1, 0, true, next, 0);
            quarter(pipe_c, f2, has_next, 0, s0, 0, s1, pend, true);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) fq1 = load_frags(next, 1, 3, true, next, 16);
            quarter(pipe_c, f3, has_next, 3, s0, 1, s1, pend, true);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                // slot (t + 3) % 4 held tile t - 1: nobody reads it after the barrier that opened this tile
                issue_tile(t + 3);
                wait_tiles_then_barrier(std::integral_constant<int, 1>{});   // own pieces of tile t + 2 (and everything older) landed
            }
        };
        using std::integral_constant;
        using std::true_type;
        // whole groups of four tiles with every constant folded; the last 1 .. 4 tiles take the run-time form of the same code
        int t = 0;
        for (; t + 4 < n_tiles; t += 4) {
            tile(t, integral_constant<int, 0>{}, true_type{});
            tile(t + 1, integral_constant<int, 1>{}, true_type{});
            tile(t + 2, integral_constant<int, 2>{}, true_type{});
            tile(t + 3, integral_constant<int, 3>{}, true_type{});
        }
        for (; t < n_tiles; ++t) tile(t, t & (kRing - 1), t + 1 < n_tiles);
        if (!kSafe) {                                            // the last quarter's P V
            Frags none;
            none.k[0] = none.k[1] = none.k[2] = u32x4{0, 0, 0, 0};
            matrix_part(none, false, 3, s0, pend, true);
        }
        l += rsum;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // trailing (unused) pieces must land before the ring is reused / released
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);     // the two lane halves hold disjoint keys of every k-step
    };

    MVI_AS3 uint32_t* const redo_flag = (MVI_AS3 uint32_t*)(lds + kLdsBytes);
    if (tid == 0) *redo_flag = 0u;                               // ordered before any read by the barriers of run()
    run(std::false_type{});
    // block-wide vote (the waves share the K / V ring and its barriers, so they repeat together or not at all)
    // (f16: P itself is packed to f16, which ends at 65504 — a row sum of at most 2^15 proves that no P of the row was larger)
    const float l_limit = std::is_same<T, __half>::value ? 0x1p15f : 0x1p100f;
    const bool out_of_range = !(l <= l_limit);                   // also true for NaN
    if (__builtin_amdgcn_ballot_w64(out_of_range) != 0ull && lane == 0) *redo_flag = 1u;
    __syncthreads();
    if (*redo_flag != 0u) {
        __syncthreads();
        run(std::true_type{});
    }

    if (qrow < Sq) {
        const float inv = 1.0f / l;
        T* op = out + ((b * Sq + qrow) * o_rs + (int64_t)h * kD);
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

}  // namespace f8

template <typename T>
static int flash8_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk, float scale, bool q_log2,
                         hipStream_t st, int64_t q_rs, int64_t kv_rs, int64_t o_rs) {
    using namespace f8;
    constexpr int kQB = 32 * kWaves;
    const int q_blocks = (Sq + kQB - 1) / kQB;
    const int64_t total = (int64_t)B * H * q_blocks;
    if (total > 0x7FFFFFFFll) return MVI_EINVAL;
    if ((int64_t)Sk * kv_rs * 2 > 0xFFFFFFFFll) return MVI_EINVAL;       // 32-bit byte offsets inside one batch entry
    // more than 64 KiB of dynamic LDS needs the opt-in attribute, once per device and instantiation
    static unsigned long long attr_set = 0ull;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MVI_EHIP;
    // see kExact. Default: exact for bf16; folded for f16, whose 11-bit mantissa makes the second rounding of Q eight times smaller
    // (below the exact bf16 form's own error) — f16 is the reference's precision recipe and the fold is worth 7 % of the kernel
    static const int fold_env = getenv("MVI_ATTN_FOLD_SCALE") ? atoi(getenv("MVI_ATTN_FOLD_SCALE")) : -1;
    // q_log2 (mvi_attention_forward_strided_qlog2): q carries scale * log2(e) from its projection's weights — the folded kernel with
    // nothing left to fold (its prologue multiplies Q by exactly 1), for both types
    const bool fold = q_log2 || (fold_env >= 0 ? fold_env != 0 : std::is_same<T, __half>::value);
    if (!((attr_set >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_flash8_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytes + 16) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_flash8_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytes + 16) != hipSuccess)
            return MVI_EHIP;
        attr_set |= 1ull << dev;
    }
    auto kern = fold ? &attn_flash8_kernel<T, false> : &attn_flash8_kernel<T, true>;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64 * kWaves), kLdsBytes + 16, st, (const T*)q, (const T*)k, (const T*)v,
                       (T*)out, H, Sq, Sk, q_log2 ? 1.0f : scale * 1.4426950408889634f, q_blocks, (int)total, q_rs, kv_rs, o_rs);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

template <typename T>
int attn_flash8_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk,
                       float scale, bool q_log2, hipStream_t st, int64_t q_rs, int64_t kv_rs, int64_t o_rs) {
    const int64_t hd = (int64_t)H * f8::kD;
    if (q_rs == 0) q_rs = hd;
    if (kv_rs == 0) kv_rs = hd;
    if (o_rs == 0) o_rs = hd;
    return flash8_launch<T>(q, k, v, out, B, H, Sq, Sk, scale, q_log2, st, q_rs, kv_rs, o_rs);
}
template int attn_flash8_launch<__hip_bfloat16>(const void*, const void*, const void*, void*, int, int, int, int, float, bool, hipStream_t, int64_t, int64_t, int64_t);
template int attn_flash8_launch<__half>(const void*, const void*, const void*, void*, int, int, int, int, float, bool, hipStream_t, int64_t, int64_t, int64_t);

}  // namespace mvi
