// EXPERIMENT (tools/attn_dev): the attention kernel with 64 query rows per wave — two 32-row query blocks share every K / V
// fragment read from LDS (half the LDS bytes per MFMA of csrc/attn_flash8.hip), 4 waves = 256 rows per block, one wave per SIMD.
// Defines the same launcher symbol as csrc/attn_flash8.hip so that tools/attn_dev/build_q64.sh can link it in its place.
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
    __device__ static u32x4 ones() { return u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; }
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
    __device__ static u32x4 ones() { return u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}; }
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
template <int N, bool kBarrier = true> __device__ __forceinline__ void wait_vm_then_barrier() {
    if (kBarrier) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");        // (timing experiment only)
}


// ---- 64 query rows per wave (two 32-row query blocks share every K / V fragment), 4 waves = 256 rows per block, ONE wave per SIMD
constexpr int kNQ = 2;
constexpr int kW = 4;
#ifndef Q64_SPREAD
#define Q64_SPREAD 1   // 1: one DMA piece per quarter; 0: all four at the end of the tile
#endif
#ifndef Q64_MSUM
#define Q64_MSUM 1     // 1: row sums on the matrix pipe; 0: v_add
#endif
#ifndef Q64_STAGE
#define Q64_STAGE 1    // 1: K / V through registers; 0: LDS-DMA
#endif
#ifndef Q64_NOLOAD
#define Q64_NOLOAD 0   // timing ablation: 1 no K / V loads inside the loop, 2 loads but no LDS writes
#endif
#ifndef Q64_X
#define Q64_X 0        // timing ablations (wrong results on purpose): 1 no exp, 2 no DMA in the loop, 3 no barrier, 4 no LDS fragment reads, 5 no softmax VALU at all
#endif

template <typename T>
__global__ __launch_bounds__(64 * kW) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_q64_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ out,
                     int H, int Sq, int Sk, float scale_log2e, int q_blocks, int total_blocks, int64_t q_rs,
                     int64_t kv_rs, int64_t o_rs) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int kQB = 32 * kNQ * kW;           // query rows per block
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
    int qrow[kNQ];
#pragma unroll
    for (int u = 0; u < kNQ; ++u) qrow[u] = qb * kQB + wave * (32 * kNQ) + 32 * u + qcol;
    const float sc_mul = scale_log2e;

    frag qf[kNQ][4];
#pragma unroll
    for (int u = 0; u < kNQ; ++u) {
        const T* qp = q + ((b * Sq + (qrow[u] < Sq ? qrow[u] : 0)) * q_rs + (int64_t)h * kD + 8 * hh);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 raw = qrow[u] < Sq ? *reinterpret_cast<const u32x4*>(qp + 16 * s) : u32x4{0, 0, 0, 0};
            qf[u][s] = as_frag<frag>(raw);
        }
    }

    const char* const kbase = reinterpret_cast<const char*>(k + (b * Sk * kv_rs + (int64_t)h * kD));
    const char* const vbase = reinterpret_cast<const char*>(v + (b * Sk * kv_rs + (int64_t)h * kD));
    constexpr int kPieces = 16 / kW;             // pieces a wave moves per tile: w, w + 4, w + 8, w + 12 (two of K, two of V)
    const uint32_t row_bytes = (uint32_t)(kv_rs * 2);
    const int pslot = lane & 7;
    int p_row[kPieces], p_chunk[kPieces];
    uint32_t p_voff[kPieces], p_dst[kPieces];
    bool p_is_v[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int pc = wave + i * kW;
        p_is_v[i] = pc >= 8;
        p_row[i] = 8 * (pc & 7) + (lane >> 3);
        p_chunk[i] = p_is_v[i] ? pslot ^ (((p_row[i] >> 1) & 1) << 2) : pslot ^ ((p_row[i] >> 1) & 7);
        p_voff[i] = (uint32_t)p_row[i] * row_bytes + 16u * p_chunk[i];
        p_dst[i] = lds0 + (p_is_v[i] ? kRing * kTileBytes : 0) + 1024u * (pc & 7);
    }
    const int n_tiles = (Sk + kKT - 1) / kKT;
    const int n_full = Sk / kKT;
    auto issue_piece = [&](int tt, int i) __attribute__((always_inline)) {
        const uint32_t ring_off = (uint32_t)((tt & (kRing - 1)) * kTileBytes);
        const char* const base = p_is_v[i] ? vbase : kbase;
        if (tt < n_full) {
            dma_piece(base + (int64_t)tt * kKT * row_bytes, p_voff[i], p_dst[i] + ring_off);
        } else {
            int r = tt * kKT + p_row[i];
            r = r < Sk ? r : Sk - 1;
            dma_piece(base, (uint32_t)r * row_bytes + 16u * p_chunk[i], p_dst[i] + ring_off);
        }
    };
    auto issue_tile = [&](int tt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) issue_piece(tt, i);
    };
    auto wait_tiles_then_barrier = [&](auto tiles_c) __attribute__((always_inline)) {
        constexpr int kT = decltype(tiles_c)::value;
        wait_vm_then_barrier<kPieces * kT>();
    };
    // Q64_STAGE: K / V tiles through registers (global_load_dwordx4 -> ds_write_b128 one tile later) instead of LDS-DMA: an LDS-DMA
    // piece costs the issuing wave ~200 cycles here and nothing else runs on its SIMD meanwhile (stamps: 4 pieces = 770 of 2500
    // cycles per tile). Same LDS image: the lane that DMA would have fed loads that chunk and writes it at its lane-linear slot.
    uint32_t p_rel[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) p_rel[i] = p_dst[i] - lds0 + 16u * (uint32_t)lane;
    u32x4 stg[2][kPieces];
    auto load_stage = [&](int tt, u32x4 (&st)[kPieces]) __attribute__((always_inline)) {
        // branch-free (rows past the end are clamped): with a branch per load the compiler's wait-count pass loses the order of the
        // loads and waits for ALL of them — this tile's included — before the first ds_write of the previous tile's registers
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const char* const base = p_is_v[i] ? vbase : kbase;
            int r = tt * kKT + p_row[i];
            r = r < Sk ? r : Sk - 1;
            st[i] = *reinterpret_cast<const u32x4*>(base + ((uint32_t)r * row_bytes + 16u * (uint32_t)p_chunk[i]));
        }
    };
    auto store_stage = [&](int tt, const u32x4 (&st)[kPieces]) __attribute__((always_inline)) {
        const uint32_t ring_off = (uint32_t)((tt & (kRing - 1)) * kTileBytes);
#pragma unroll
        for (int i = 0; i < kPieces; ++i) *reinterpret_cast<MVI_AS3 u32x4*>(lds + p_rel[i] + ring_off) = st[i];
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    uint32_t ka[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = (uint32_t)(qcol * 128 + (((2 * s + hh) ^ ((qcol >> 1) & 7)) << 4));
    uint32_t va[2];
    {
        const int i16 = lane & 15, qq = i16 >> 2, p = i16 & 3, g1 = (lane >> 4) & 1, sel = (qq >> 1) & 1;
        const uint32_t base = (uint32_t)((4 * hh + qq) * 128 + ((2 * g1 + (p >> 1)) << 4) + 8 * (p & 1));
        va[0] = kRing * kTileBytes + base + 64u * sel;
        va[1] = kRing * kTileBytes + base + 64u * (1 - sel);
    }

    f32x16 o[kNQ][2], s0[kNQ], s1[kNQ], ls[kNQ];   // ls: row sums of P as an MFMA against a fragment of ones (every row of the tile holds them)
    float nm[kNQ], l[kNQ];            // nm = -m * scale * log2 e: P = exp2(s * sc_mul + nm), one fma per score (the multiply the exact form needs anyway)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto qk2 = [&](int slot, int kb, int s_lo, f32x16 (&sc)[kNQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = s_lo; s < s_lo + 2; ++s) {
            const u32x4 kf = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ka[s] + slot * kTileBytes + kb * 4096);
#pragma unroll
            for (int u = 0; u < kNQ; ++u) sc[u] = M::mfma(as_frag<frag>(kf), qf[u][s], s == 0 ? zero16 : sc[u]);
        }
    };
    auto block_max = [&](const f32x16& sc) __attribute__((always_inline)) {
        float ra = __builtin_fmaxf(sc[0], sc[1]), rb = __builtin_fmaxf(sc[2], sc[3]);
#pragma unroll
        for (int r = 4; r < 16; r += 4) {
            ra = __builtin_fmaxf(__builtin_fmaxf(ra, sc[r]), sc[r + 1]);
            rb = __builtin_fmaxf(__builtin_fmaxf(rb, sc[r + 2]), sc[r + 3]);
        }
        const float rmax = __builtin_fmaxf(ra, rb);
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
        return __builtin_fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    // moves the reference exponent of the lanes in `grow` to their block's row max (in exp2 units: delta2 = rmax * sc_mul + nm)
    auto rescale = [&](int u, bool grow, float rmax, bool first) __attribute__((always_inline)) {
        const float delta2 = grow ? __builtin_fmaf(rmax, sc_mul, nm[u]) : 0.f;
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta2);
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[u][0][i] *= alpha; o[u][1][i] *= alpha; ls[u][i] *= alpha; }
        nm[u] -= delta2;
    };
    struct Frags { u32x4 k[2]; u32x2 v[2][2]; };
    auto load_frags = [&](int kslot, int kb, int s_lo, bool with_k, int vslot, int vkeys) __attribute__((always_inline)) {
        Frags f;
        if (Q64_X == 4) {
            f.k[0] = f.k[1] = u32x4{0x3c003c00u, 0, 0, 0};
            f.v[0][0] = f.v[0][1] = f.v[1][0] = f.v[1][1] = u32x2{0x3c003c00u, 0};
            return f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            f.k[i] = with_k ? *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ka[s_lo + i] + kslot * kTileBytes + kb * 4096) : u32x4{0, 0, 0, 0};
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
    auto probs = [&](int u, int s2, const f32x16& sc) __attribute__((always_inline)) {
        u32x4 pr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = __builtin_fmaf(sc[8 * s2 + 2 * i], sc_mul, nm[u]), a1 = __builtin_fmaf(sc[8 * s2 + 2 * i + 1], sc_mul, nm[u]);
            if (Q64_X == 5) { pr[i] = __float_as_uint(sc[8 * s2 + 2 * i]); continue; }
            const float p0 = Q64_X == 1 ? a0 : __builtin_amdgcn_exp2f(a0);
            const float p1 = Q64_X == 1 ? a1 : __builtin_amdgcn_exp2f(a1);
            if (!Q64_MSUM) ls[u][0] += p0 + p1;
            pr[i] = M::pack2(p0, p1);
        }
        if (!Q64_MSUM) asm volatile("" : "+v"(ls[u][0]));
        return pr;
    };
    struct Pending { u32x4 p[kNQ]; u32x2 v[2][2]; };
    auto matrix_part = [&](const Frags& f, bool with_k, int s_lo, f32x16 (&acc)[kNQ], const Pending& pd, bool with_pv) __attribute__((always_inline)) {
        const u32x4 av0 = {pd.v[0][0][0], pd.v[0][0][1], pd.v[0][1][0], pd.v[0][1][1]};
        const u32x4 av1 = {pd.v[1][0][0], pd.v[1][0][1], pd.v[1][1][0], pd.v[1][1][1]};
        // every fragment feeds both query blocks back to back (two independent accumulator chains); a block's S' chain is two
        // d-steps per quarter; the row sums of P ride on the matrix pipe (P x ones) instead of 16 v_add per quarter and block
        const frag one = as_frag<frag>(M::ones());
        const int s = s_lo;
#pragma unroll
        for (int u = 0; u < kNQ; ++u) if (with_k) acc[u] = M::mfma(as_frag<frag>(f.k[0]), qf[u][s], s == 0 ? zero16 : acc[u]);
#pragma unroll
        for (int u = 0; u < kNQ; ++u) if (with_pv) o[u][0] = M::mfma(as_frag<frag>(av0), as_frag<frag>(pd.p[u]), o[u][0]);
#pragma unroll
        for (int u = 0; u < kNQ; ++u) if (with_k) acc[u] = M::mfma(as_frag<frag>(f.k[1]), qf[u][s + 1], acc[u]);
#pragma unroll
        for (int u = 0; u < kNQ; ++u) if (with_pv) o[u][1] = M::mfma(as_frag<frag>(av1), as_frag<frag>(pd.p[u]), o[u][1]);
#pragma unroll
        for (int u = 0; u < kNQ; ++u) if (with_pv && Q64_MSUM) ls[u] = M::mfma(one, as_frag<frag>(pd.p[u]), ls[u]);
    };
    auto quarter = [&](auto pipelined_c, const Frags& f, bool with_k, int s_lo, f32x16 (&acc)[kNQ], int s2, const f32x16 (&sc)[kNQ],
                       Pending& pend, bool pend_valid) __attribute__((always_inline)) {
        constexpr bool kPipe = decltype(pipelined_c)::value;
        if (kPipe) {
            matrix_part(f, with_k, s_lo, acc, pend, pend_valid);
            Pending nx;
#pragma unroll
            for (int u = 0; u < kNQ; ++u) nx.p[u] = probs(u, s2, sc[u]);
            nx.v[0][0] = f.v[0][0]; nx.v[0][1] = f.v[0][1]; nx.v[1][0] = f.v[1][0]; nx.v[1][1] = f.v[1][1];
            pend = nx;
        } else {
            Pending now;
#pragma unroll
            for (int u = 0; u < kNQ; ++u) now.p[u] = probs(u, s2, sc[u]);
            now.v[0][0] = f.v[0][0]; now.v[0][1] = f.v[0][1]; now.v[1][0] = f.v[1][0]; now.v[1][1] = f.v[1][1];
            matrix_part(f, with_k, s_lo, acc, now, true);
        }
    };

    auto run = [&](auto safe_c) __attribute__((always_inline)) {
        constexpr bool kSafe = decltype(safe_c)::value;
#pragma unroll
        for (int u = 0; u < kNQ; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[u][0][i] = 0.f; o[u][1][i] = 0.f; ls[u][i] = 0.f; }
            l[u] = 0.f; nm[u] = 0.f;
        }
        if (Q64_STAGE) {
            load_stage(0, stg[0]);
            if (n_tiles > 1) load_stage(1, stg[1]);
            store_stage(0, stg[0]);
            if (n_tiles > 1) store_stage(1, stg[1]);
            if (n_tiles > 2) load_stage(2, stg[0]);          // written to LDS during tile 0
            lds_barrier();
        } else {
            issue_tile(0);
            issue_tile(1);
            issue_tile(2);
            wait_tiles_then_barrier(std::integral_constant<int, 2>{});
        }
        qk2(0, 0, 0, s0);
        qk2(0, 0, 2, s0);
        if (Sk < 32) {
#pragma unroll
            for (int u = 0; u < kNQ; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (((r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) s0[u][r] = -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < kNQ; ++u) rescale(u, true, block_max(s0[u]), true);
        if (!Q64_STAGE) wait_tiles_then_barrier(std::integral_constant<int, 1>{});
        Frags fq0 = load_frags(0, 1, 0, true, 0, 0), fq1 = load_frags(0, 1, 2, true, 0, 16);

        auto decide = [&](const f32x16 (&sc)[kNQ]) __attribute__((always_inline)) {
            if (!kSafe) return;
#pragma unroll
            for (int u = 0; u < kNQ; ++u) {
                const float rmax = block_max(sc[u]);
                const bool grow = __builtin_fmaf(rmax, sc_mul, nm[u]) > kRescaleThreshold;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(grow) != 0ull, 0)) rescale(u, grow, rmax, false);
            }
        };
        Pending pend;
#pragma unroll
        for (int u = 0; u < kNQ; ++u) pend.p[u] = u32x4{0, 0, 0, 0};
        pend.v[0][0] = pend.v[0][1] = pend.v[1][0] = pend.v[1][1] = u32x2{0, 0};
        constexpr std::integral_constant<bool, !kSafe> pipe_c{};
        uint64_t stamp[6] = {0, 0, 0, 0, 0, 0};
        const uint64_t clk0 = Q64_X == 9 ? __builtin_amdgcn_s_memtime() : 0, rt0 = Q64_X == 9 ? __builtin_amdgcn_s_memrealtime() : 0;
        auto tile = [&](int t, auto slot_c, auto has_next_c) __attribute__((always_inline)) {
            const int slot = slot_c, next = (slot + 1) & (kRing - 1);
            uint64_t ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
            if (Q64_X == 9) ts0 = __builtin_amdgcn_s_memtime();
            const bool has_next = has_next_c;
            const int k0 = t * kKT;
            const bool ragged = !has_next && k0 + kKT > Sk;
            if (ragged && t > 0) {
#pragma unroll
                for (int u = 0; u < kNQ; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((k0 + (r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) s0[u][r] = -INFINITY;
            }
            if (Q64_STAGE && Q64_X != 2 && Q64_NOLOAD != 1) load_stage(t + 3, stg[(slot + 1) & 1]);        // (past the end: clamped rows, never written to LDS)
            decide(s0);
            Frags f2 = load_frags(next, 0, 0, has_next, slot, 32);
            if (!Q64_STAGE && has_next && Q64_X != 2 && Q64_SPREAD) issue_piece(t + 3, 0);
            quarter(pipe_c, fq0, true, 0, s1, 0, s0, pend, t > 0);
            __builtin_amdgcn_sched_barrier(0);
            if (Q64_X == 9) ts1 = __builtin_amdgcn_s_memtime();
            Frags f3 = load_frags(next, 0, 2, has_next, slot, 48);
            if (!Q64_STAGE && has_next && Q64_X != 2 && Q64_SPREAD) issue_piece(t + 3, 1);
            quarter(pipe_c, fq1, true, 2, s1, 1, s0, pend, true);
            __builtin_amdgcn_sched_barrier(0);
            if (Q64_X == 9) ts2 = __builtin_amdgcn_s_memtime();
            if (ragged) {
#pragma unroll
                for (int u = 0; u < kNQ; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((k0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) >= Sk) s1[u][r] = -INFINITY;
            }
            decide(s1);
            if (has_next) fq0 = load_frags(next, 1, 0, true, next, 0);
            if (!Q64_STAGE && has_next && Q64_X != 2 && Q64_SPREAD) issue_piece(t + 3, 2);
            quarter(pipe_c, f2, has_next, 0, s0, 0, s1, pend, true);
            __builtin_amdgcn_sched_barrier(0);
            if (Q64_X == 9) ts3 = __builtin_amdgcn_s_memtime();
            if (has_next) fq1 = load_frags(next, 1, 2, true, next, 16);
            if (!Q64_STAGE && has_next && Q64_X != 2 && Q64_SPREAD) issue_piece(t + 3, 3);
            quarter(pipe_c, f3, has_next, 2, s0, 1, s1, pend, true);
            __builtin_amdgcn_sched_barrier(0);
            if (Q64_X == 9) { ts4 = __builtin_amdgcn_s_memtime(); stamp[0] += ts1 - ts0; stamp[1] += ts2 - ts1; stamp[2] += ts3 - ts2; stamp[3] += ts4 - ts3; }
            if (Q64_STAGE) {
                if (t + 2 < n_tiles && Q64_X != 2 && Q64_NOLOAD == 0) store_stage(t + 2, stg[slot & 1]);
                if (Q64_NOLOAD == 2) asm volatile("" :: "v"(stg[slot & 1][0]), "v"(stg[slot & 1][1]), "v"(stg[slot & 1][2]), "v"(stg[slot & 1][3]));
                if (has_next) lds_barrier();
                if (Q64_X == 9 && has_next) { const uint64_t tb = __builtin_amdgcn_s_memtime(); stamp[4] += tb - ts4; stamp[5] += 1; }
            } else if (has_next) {
                if (Q64_X != 2 && !Q64_SPREAD) issue_tile(t + 3);
                if (Q64_X == 2) asm volatile("s_barrier" ::: "memory");
                else if (Q64_X == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces) : "memory");
                else wait_tiles_then_barrier(std::integral_constant<int, 1>{});
                if (Q64_X == 9) { const uint64_t tb = __builtin_amdgcn_s_memtime(); stamp[4] += tb - ts4; stamp[5] += 1; }
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
        // the last 1 .. 4 tiles: t is a multiple of 4 here, so the slots (and the staging registers' parity) stay compile-time
        if (t < n_tiles) tile(t, integral_constant<int, 0>{}, t + 1 < n_tiles);
        if (t + 1 < n_tiles) tile(t + 1, integral_constant<int, 1>{}, t + 2 < n_tiles);
        if (t + 2 < n_tiles) tile(t + 2, integral_constant<int, 2>{}, t + 3 < n_tiles);
        if (t + 3 < n_tiles) tile(t + 3, integral_constant<int, 3>{}, false);
        if (Q64_X == 9 && !kSafe && lane == 0 && blockIdx.x == 7)
            printf("block 7 wave %d: %llu tiles; cycles per tile: q0 %llu q1 %llu q2 %llu q3 %llu dma+wait+barrier %llu; in-kernel clock %.0f MHz\n", wave,
                   (unsigned long long)stamp[5], (unsigned long long)(stamp[0] / stamp[5]), (unsigned long long)(stamp[1] / stamp[5]),
                   (unsigned long long)(stamp[2] / stamp[5]), (unsigned long long)(stamp[3] / stamp[5]), (unsigned long long)(stamp[4] / stamp[5]),
                   100.0 * (double)(__builtin_amdgcn_s_memtime() - clk0) / (double)(__builtin_amdgcn_s_memrealtime() - rt0));
        if (!kSafe) {
            Frags none;
            none.k[0] = none.k[1] = u32x4{0, 0, 0, 0};
            matrix_part(none, false, 2, s0, pend, true);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < kNQ; ++u) {
            l[u] = ls[u][0];                                     // the MFMA summed over all 16 keys of every fragment (both lane halves)
            if (!Q64_MSUM) {
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l[u]), __float_as_uint(l[u]), false, false);
                l[u] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
        }
    };

    MVI_AS3 uint32_t* const redo_flag = (MVI_AS3 uint32_t*)(lds + kLdsBytes);
    if (tid == 0) *redo_flag = 0u;
    run(std::false_type{});
    bool out_of_range = false;
#pragma unroll
    for (int u = 0; u < kNQ; ++u) out_of_range |= !(l[u] <= 0x1p100f);
    if (__builtin_amdgcn_ballot_w64(out_of_range) != 0ull && lane == 0) *redo_flag = 1u;
    __syncthreads();
    if (*redo_flag != 0u) {
        __syncthreads();
        run(std::true_type{});
    }
#pragma unroll
    for (int u = 0; u < kNQ; ++u) {
        if (qrow[u] < Sq) {
            const float inv = 1.0f / l[u];
            T* op = out + ((b * Sq + qrow[u]) * o_rs + (int64_t)h * kD);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w = {M::pack2(o[u][db][4 * g] * inv, o[u][db][4 * g + 1] * inv),
                               M::pack2(o[u][db][4 * g + 2] * inv, o[u][db][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2*>(op + 32 * db + 8 * g + 4 * hh) = w;
                }
        }
    }
}

}  // namespace f8

template <typename T>
int attn_flash8_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Sk,
                       float scale, hipStream_t st, int64_t q_rs, int64_t kv_rs, int64_t o_rs) {
    using namespace f8;
    const int64_t hd = (int64_t)H * kD;
    if (q_rs == 0) q_rs = hd;
    if (kv_rs == 0) kv_rs = hd;
    if (o_rs == 0) o_rs = hd;
    constexpr int kQB = 32 * kNQ * kW;
    const int q_blocks = (Sq + kQB - 1) / kQB;
    const int64_t total = (int64_t)B * H * q_blocks;
    if (total > 0x7FFFFFFFll) return MVI_EINVAL;
    if ((int64_t)Sk * kv_rs * 2 > 0xFFFFFFFFll) return MVI_EINVAL;
    auto kern = &attn_q64_kernel<T>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes + 16);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64 * kW), kLdsBytes + 16, st, (const T*)q, (const T*)k, (const T*)v,
                       (T*)out, H, Sq, Sk, scale * 1.4426950408889634f, q_blocks, (int)total, q_rs, kv_rs, o_rs);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}
template int attn_flash8_launch<__hip_bfloat16>(const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t, int64_t, int64_t, int64_t);
template int attn_flash8_launch<__half>(const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t, int64_t, int64_t, int64_t);

}  // namespace mvi
