// Linear with 320 outputs and a long contraction (K = 1280: the second projection of the 21 level-0 FeedForward layers and their
// temporal twins, svd_inpaint1/sgm/modules/attention.py:98-115 `net[2]`) for gfx950, bf16 / f16 MFMA:
//     out[r, n] = x[r, :] . W[n, :] + b[n],   r < rows, n < 320, K a multiple of 64
// Why it exists: at [258048, 1280] x [1280, 320] the library's 256 x 256 macro-tile covers 320 columns with two column tiles, 37 % of
// the second one empty, and runs at 0.60 PFLOP/s (0.35 ms per call, 42 calls = 14.9 ms of a 168 ms step). Here a block's 256 rows
// keep ALL 320 outputs in accumulators (a wave: 32 rows x 10 column tiles = 160 registers), so nothing is padded and x is read once.
//
// Structure (csrc/ff_geglu.hip with the roles turned: there x is stationary and the outputs stream, here the outputs are stationary
// and K streams):
//   * block = 8 waves x 32 rows; K advances in chunks of 64; per chunk and wave 4 k-steps x 10 column tiles = 40 MFMAs
//     (v_mfma_f32_32x32x16, A = x rows, B = W rows: the output column sits on the lane);
//   * W chunk [320 rows][64 k] = 40 KiB streams through a 3-slot LDS ring by LDS-DMA (128-byte rows, 16-byte chunk c of row r at
//     slot c ^ ((r >> 1) & 7), swizzled on the source side: the K image of attn_flash8.hip, conflict-free ds_read_b128); all 8 waves
//     read the same chunk; 4 loader waves issue the chunk two ahead at the END of a chunk and wait with a counted vmcnt;
//   * a wave's own x rows go HBM -> registers directly (4 x global_load_dwordx4 per chunk, one chunk ahead, double-buffered;
//     the 4 loads of a chunk touch the same 32 lines). They are inline assembly like the DMA: a load the compiler knows about
//     makes it wait for vmcnt(0) at the first use — the DMA pieces just issued included;
//   * the bias is the accumulators' initial value; outputs leave through a wave-private 4 KiB LDS tile as 16-byte stores
//     (two column tiles = one 128-byte line per row per flush) into a buffer padded to whole 256-row blocks.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {
int unet_fail(int code, const char* msg);
namespace ln3 {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define MVI_AS3 __attribute__((address_space(3)))

constexpr int kN = 320;                              // outputs
constexpr int kNT = kN / 32;                         // column tiles per wave
constexpr int kWaves = 8;
constexpr int kRows = 32 * kWaves;                   // x rows per block
constexpr int kKC = 64;                              // contraction elements per chunk
constexpr int kChunkBytes = kN * kKC * 2;            // 40960
constexpr int kPieces = kChunkBytes / 1024;          // 40 LDS-DMA pieces of 1 KiB (8 W rows each)
constexpr int kRing = 3;
#ifndef LN3_LOADERS
#define LN3_LOADERS 4
#endif
#ifndef LN3_AHEAD
#define LN3_AHEAD 4
#endif
#ifndef LN3_SGB
#define LN3_SGB 1
#endif
constexpr int kLoaders = LN3_LOADERS;
constexpr int kPiecesPerLoader = kPieces / kLoaders; // 10
constexpr int kLdsBytes = kRing * kChunkBytes + kWaves * 4096;

template <typename T> struct Mma;
template <> struct Mma<__hip_bfloat16> {
    using frag = bf16x8;
    __device__ static f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {
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

__device__ __forceinline__ void dma_piece(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory");
}
// 16 bytes per lane, invisible to the compiler's wait-count bookkeeping (see the header): whoever reads the result waits first
__device__ __forceinline__ u32x4 load16_async(const void* sbase, uint32_t voff) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}

// kConv: the same kernel as a 3x3 / stride 1 / padding 1 convolution of token-major (NHWC) activations — an implicit GEMM with
// K = 9 C_in ordered (tap, channel): x row r is pixel r of [N, H, W], chunk c = (tap c / cpc, channels 64 (c % cpc) ..), and the
// lane's load address moves by the tap's pixel offset. A lane whose tap falls outside the image loads its own pixel instead (a
// valid address) and the fragment is zeroed before the MFMAs (a wave-uniform branch: interior waves skip it).
// taps = 3 is the same thing with the taps along H only ((3,1,1) over frames: H = T frames, W = pixels of a frame).
// groups > 1: C_out = 320 groups; a block computes 320 of them (blocks of one row block are neighbours and share x through L2).
// kSplit (small images: fewer than half the CUs would get a block): ksplit blocks share a (row block, column group), each over a
// contiguous range of K chunks, and store fp32 partial sums [ksplit][padded rows][C_out]; splitk_reduce_kernel adds them and the bias.
// stride 2 (Downsample.op, openaimodel.py:150-166): rows are OUTPUT pixels [N, Ho, Wo]; the lane's centre is input pixel
// (2 yo, 2 xo) of the H x W input image, the taps' offsets and border tests are those of the input image.
struct ConvGeom {
    int H, W, cpc, taps, groups, ksplit;             // INPUT image height / width, 64-channel chunks per tap (C_in / 64), 9 or 3, C_out / 320
    int stride, Ho, Wo;                              // 1 or 2 (3x3 only); output height / width (= H, W at stride 1)
};

template <typename T, bool kConv, bool kSplit = false>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void linear_n320_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                        int64_t rows, int K, int64_t x_rs, int64_t o_rs, int n_blocks, ConvGeom cg, float* __restrict__ part) {
    using M = Mma<T>;
    using frag = typename M::frag;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    MVI_AS3 char* const lds = (MVI_AS3 char*)smem;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;
    int bid = blockIdx.x;
    if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // neighbouring row blocks on one XCD (W is shared by all)
    int part_col0 = 0, ks = 0;
    if (kConv && cg.groups > 1) {                    // (row block, column group): the groups of one row block run side by side
        const int g = bid % cg.groups;
        bid /= cg.groups;
        w += (int64_t)g * kN * K;
        out += g * kN;
        part_col0 = g * kN;
        if (bias) bias += g * kN;
    }
    if (kSplit) {
        ks = bid % cg.ksplit;
        bid /= cg.ksplit;
    }
    const int64_t row0 = (int64_t)bid * kRows + wave * 32;                       // wave-uniform
    const int64_t row = row0 + col;
    // this block's chunks: c0 .. c0 + n_chunks - 1 of the K / 64 (all of them unless kSplit); chunk indices below are relative to c0
    const int c0 = kSplit ? (int)((int64_t)ks * (K / kKC) / cg.ksplit) : 0;
    const int n_chunks = kSplit ? (int)((int64_t)(ks + 1) * (K / kKC) / cg.ksplit) - c0 : K / kKC;

    // ---- accumulators start from the bias: column 32 j + col of every row this lane holds
    f32x16 acc[kNT];
#pragma unroll
    for (int j = 0; j < kNT; ++j) {
        const float b = bias && !kSplit ? bias[32 * j + col] : 0.f;      // (kSplit: the reduction adds it)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = b;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the compiler's own loads are done before the hand-counted ones start

    // ---- x: A operand, element e of lane (col, hh), k-step s of chunk c: x[row][64 c + 16 s + 8 hh + e]; rows past the end read the last row
    // (kConv: the base is the tensor and the lane offset absolute — a tap's row may lie before the wave's first one)
    const char* const xbase = kConv ? reinterpret_cast<const char*>(x)
                                    : reinterpret_cast<const char*>(x + (row0 < rows ? row0 : rows - 1) * x_rs);   // wave-uniform
    const int64_t rclamp = kConv ? (row < rows ? row : rows - 1)
                                 : (row < rows ? (row - (row0 < rows ? row0 : rows - 1)) : 0);             // (a block never starts past the end)
    uint32_t x_voff = (uint32_t)(rclamp * x_rs * 2 + 16 * hh);
    uint32_t tap_ok = 0x1FFu;                        // bit 3 (dy + 1) + (dx + 1): that neighbour of the lane's pixel is inside the image
    if (kConv) {
        const int64_t img = rclamp / ((int64_t)cg.Ho * cg.Wo);
        const int pix = (int)(rclamp - img * ((int64_t)cg.Ho * cg.Wo));
        const int py = pix / cg.Wo * cg.stride, px = (pix - pix / cg.Wo * cg.Wo) * cg.stride;      // the centre, in the input image
        if (cg.stride != 1) x_voff = (uint32_t)(((img * cg.H + py) * cg.W + px) * x_rs * 2 + 16 * hh);
        if (cg.taps == 9) {
            if (py == 0) tap_ok &= ~0x007u;
            if (py == cg.H - 1) tap_ok &= ~0x1C0u;
            if (px == 0) tap_ok &= ~0x049u;
            if (px == cg.W - 1) tap_ok &= ~0x124u;
        } else {                                     // bit dy + 1
            if (py == 0) tap_ok &= ~0x1u;
            if (py == cg.H - 1) tap_ok &= ~0x4u;
        }
    }
    int ld_tap = kConv ? c0 / cg.cpc : 0, ld_cc = kConv ? c0 - ld_tap * cg.cpc : 0;   // kConv: (tap, channel chunk) of the next load_x call — they come in chunk order
    // returns the lane's keep mask for the fragment (all ones unless kConv and the tap is outside the image)
    auto load_x = [&](int c, u32x4 (&xr)[4]) __attribute__((always_inline)) -> uint32_t {
        if (!kConv) {
            const char* const base = xbase + (int64_t)(c0 + c) * (kKC * 2);
#pragma unroll
            for (int s = 0; s < 4; ++s) xr[s] = load16_async(base + 32 * s, x_voff);
            return ~0u;
        }
        const int dy = cg.taps == 9 ? ld_tap / 3 - 1 : ld_tap - 1, dx = cg.taps == 9 ? ld_tap - 3 * (ld_tap / 3) - 1 : 0;
        const int delta = (dy * cg.W + dx) * (int)(x_rs * 2);
        const bool ok = (tap_ok >> ld_tap) & 1u;
        const uint32_t voff = ok ? x_voff + (uint32_t)delta : x_voff;
        const char* const base = xbase + ld_cc * (kKC * 2);
#pragma unroll
        for (int s = 0; s < 4; ++s) xr[s] = load16_async(base + 32 * s, voff);
        if (c + 1 < n_chunks) {                      // (the calls past the end repeat the last chunk)
            if (++ld_cc == cg.cpc) { ld_cc = 0; ++ld_tap; }
        }
        return ok ? ~0u : 0u;
    };

    // ---- LDS-DMA source addressing: piece p = W rows 8 p .. 8 p + 7 of the chunk (128 bytes each); lane i fills (row 8 p + i / 8,
    // slot i % 8) with source chunk slot ^ ((row >> 1) & 7)
    const bool loader = wave < kLoaders;
    uint32_t p_voff[kPiecesPerLoader];
    const uint32_t w_row_bytes = (uint32_t)K * 2u;
#pragma unroll
    for (int i = 0; i < kPiecesPerLoader; ++i) {
        const uint32_t pc = (uint32_t)(wave + i * kLoaders);         // (meaningless for a wave that loads nothing)
        const uint32_t r = 8u * pc + (uint32_t)(lane >> 3), slot = (uint32_t)(lane & 7);
        p_voff[i] = r * w_row_bytes + 16u * (slot ^ ((r >> 1) & 7u));
    }
    const char* const wbase = reinterpret_cast<const char*>(w) + (int64_t)c0 * (kKC * 2);
    auto issue_chunk = [&](int c) __attribute__((always_inline)) {
        // chunks past the end re-load the last one (never read): every call issues the same number of pieces, the counted wait stays valid
        const int cc = c < n_chunks ? c : n_chunks - 1;
        const uint32_t slot_off = (uint32_t)((c % kRing) * kChunkBytes);
        const char* const base = wbase + (int64_t)cc * (kKC * 2);
#pragma unroll
        for (int i = 0; i < kPiecesPerLoader; ++i) dma_piece(base, p_voff[i], lds0 + slot_off + 1024u * (uint32_t)(wave + i * kLoaders));
    };

    // ---- LDS read addressing: B operand = W rows; lane (col, hh), column tile j, k-step s reads row 32 j + col, chunk 2 s + hh
    uint32_t ka[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = (uint32_t)(col * 128 + (((2 * s + hh) ^ ((col >> 1) & 7)) << 4));
    auto wfrag = [&](uint32_t slot_base, int q) __attribute__((always_inline)) {      // q = 10 s + j
        const int s = q / kNT, j = q % kNT;
        return *reinterpret_cast<MVI_AS3 const u32x4*>(lds + slot_base + ka[s] + j * 4096);
    };

    // One chunk: 40 MFMAs, W fragments requested kAhead ahead. Ten independent accumulator chains: no MFMA waits for the one before it.
    constexpr int kAhead = LN3_AHEAD;
    auto chunk_fn = [&](uint32_t slot_base, u32x4 (&xr)[4], uint32_t keep) __attribute__((always_inline)) {
        if (kConv && __builtin_amdgcn_ballot_w64(keep == 0u) != 0ull) {
            // the fragment is an output of assembly the compiler believes complete: this statement (volatile, so it stays behind the
            // wait that closed the previous chunk) is what the masking depends on
            asm volatile("" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
#pragma unroll
            for (int s = 0; s < 4; ++s) xr[s] &= keep;
        }
        u32x4 wf[kAhead + 1];
#pragma unroll
        for (int q = 0; q < kAhead; ++q) wf[q] = wfrag(slot_base, q);
#pragma unroll
        for (int q = 0; q < 4 * kNT; ++q) {
            if (q + kAhead < 4 * kNT) wf[(q + kAhead) % (kAhead + 1)] = wfrag(slot_base, q + kAhead);
            acc[q % kNT] = M::mfma(as_frag<frag>(xr[q / kNT]), as_frag<frag>(wf[q % (kAhead + 1)]), acc[q % kNT]);
        }
        // the order above is the order wanted: left alone, the scheduler sinks every read to just before its MFMA (ds_read, wait
        // lgkmcnt(0), MFMA, 40 times per chunk) and the LDS latency is exposed: 1.13-1.24 PFLOP/s instead of 1.23-1.32
#if LN3_SGB
        __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
        for (int q = 0; q < 4 * kNT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q + kAhead < 4 * kNT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#endif
    };
    // Loaders issue the chunk two ahead (its slot held chunk c - 1, which nobody reads any more), then every wave waits for
    // everything older than those pieces — the next chunk's W pieces and its own next x rows among them — and the block meets.
    // (The x registers are NOT operands of the wait: tied operands made the allocator copy them in front of it, i.e. before the
    // loads had landed. Nothing that uses them can move above the wait anyway: every MFMA also takes a W fragment read from LDS
    // after it, and LDS reads do not cross a statement that clobbers memory.)
    auto close_chunk = [&](int c) __attribute__((always_inline)) {
        if (loader) {
            issue_chunk(c + 2);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kPiecesPerLoader) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: W chunks 0 and 1 in flight, x of chunk 0; chunk 0 landed
    u32x4 xa[4], xb[4];
    uint32_t ka_keep = load_x(0, xa), kb_keep = ~0u;
    if (loader) {
        issue_chunk(0);
        issue_chunk(1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kPiecesPerLoader) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);

    auto next_slot = [&](uint32_t s) __attribute__((always_inline)) { return s + kChunkBytes == (uint32_t)(kRing * kChunkBytes) ? 0u : s + kChunkBytes; };
    uint32_t slot = 0;
    int c = 0;
    for (; c + 1 < n_chunks; c += 2) {
        kb_keep = load_x(c + 1, xb);                                 // (c + 1 < n_chunks)
        chunk_fn(slot, xa, ka_keep);
        close_chunk(c);
        slot = next_slot(slot);
        ka_keep = load_x(c + 2 < n_chunks ? c + 2 : n_chunks - 1, xa);   // past the end: re-read the last chunk's rows (never used)
        chunk_fn(slot, xb, kb_keep);
        close_chunk(c + 1);
        slot = next_slot(slot);
    }
    if (c < n_chunks) chunk_fn(slot, xa, ka_keep);                   // K / 64 odd: one chunk left
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // trailing (unused) pieces and rows land before the block ends

    if (kSplit) {
        // fp32 partial sums straight from the accumulators: register i of column tile j = row (i & 3) + 8 (i >> 2) + 4 hh, column 32 j + col
        const int64_t c_tot = (int64_t)cg.groups * kN, rows_pad = (int64_t)(n_blocks / (cg.groups * cg.ksplit)) * kRows;
        float* const pbase = part + ((int64_t)ks * rows_pad + row0 + 4 * hh) * c_tot + part_col0 + col;
#pragma unroll
        for (int j = 0; j < kNT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) pbase[((i & 3) + 8 * (i >> 2)) * c_tot + 32 * j] = acc[j][i];
        return;
    }
    // ---- outputs: two column tiles at a time through the wave's LDS tile [32 rows][64 columns], then four 16-byte stores per lane
    char* const obase = reinterpret_cast<char*>(out + row0 * o_rs);
    const int64_t orow_bytes = o_rs * 2;
    const uint32_t otile = (uint32_t)(kRing * kChunkBytes + wave * 4096);
    const uint32_t ot_w = otile + (uint32_t)(4 * hh * 128 + col * 2);              // + 128 * row of the register (+ 64 for the second tile)
    const uint32_t ot_r = otile + (uint32_t)((lane >> 3) * 128 + (lane & 7) * 16);  // + 1024 i: rows 8 i + lane / 8
    const int64_t st_off = (int64_t)(lane >> 3) * orow_bytes + (lane & 7) * 16;
#pragma unroll
    for (int j = 0; j < kNT; j += 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t pk = M::pack2(acc[j + t][r], acc[j + t][r + 1]);
                const int m = (r & 3) + 8 * (r >> 2);
                *reinterpret_cast<MVI_AS3 uint16_t*>(lds + ot_w + 64 * t + m * 128) = (uint16_t)(pk & 0xFFFFu);
                *reinterpret_cast<MVI_AS3 uint16_t*>(lds + ot_w + 64 * t + (m + 1) * 128) = (uint16_t)(pk >> 16);
            }
        }
        char* const op = obase + (j / 2) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ot_r + 1024 * i);
            *reinterpret_cast<u32x4*>(op + (8 * i) * orow_bytes + st_off) = v;
        }
    }
}

// out[r][c] = sum_s part[s][r][c] + bias[c], eight columns per thread
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, T* __restrict__ out,
                                                            int64_t rows, int c_tot, int64_t o_rs, int ksplit, int64_t rows_pad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int vpr = c_tot / 8;
    if (i >= rows * vpr) return;
    const int64_t r = i / vpr;
    const int c = (int)(i - r * vpr) * 8;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = bias ? bias[c + k] : 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const float4* const p = reinterpret_cast<const float4*>(part + ((int64_t)s * rows_pad + r) * c_tot + c);
        const float4 u = p[0], v = p[1];
        a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w; a[4] += v.x; a[5] += v.y; a[6] += v.z; a[7] += v.w;
    }
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = Mma<T>::pack2(a[2 * k], a[2 * k + 1]);
    *reinterpret_cast<u32x4*>(out + r * o_rs + c) = o;
}

}  // namespace ln3

template <typename T, bool kConv = false, bool kSplit = false>
static int linear_n320_launch(const void* x, const void* w, const float* bias, void* out, int64_t rows, int K, int64_t x_rs, int64_t o_rs,
                              hipStream_t st, ln3::ConvGeom cg = {0, 0, 0, 0, 1, 1, 1, 0, 0}, float* part = nullptr) {
    using namespace ln3;
    const int64_t n_blocks = (rows + kRows - 1) / kRows * cg.groups * cg.ksplit;
    if (n_blocks > 0x7FFFFFFFll) return MVI_EINVAL;
    static unsigned long long attr_set = 0;                      // per device and instantiation: the opt-in for > 64 KiB of dynamic LDS
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MVI_EHIP;
    auto kern = &linear_n320_kernel<T, kConv, kSplit>;
    if (!(attr_set >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MVI_EHIP;
        attr_set |= 1ull << dev;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks), dim3(64 * kWaves), kLdsBytes, st, (const T*)x, (const T*)w, bias, (T*)out, rows, K, x_rs,
                       o_rs, (int)n_blocks, cg, part);
    if (kSplit) {
        const int c_tot = cg.groups * kN;
        const int64_t threads = rows * (c_tot / 8), rows_pad = (rows + kRows - 1) / kRows * kRows;
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, (T*)out,
                           rows, c_tot, o_rs, cg.ksplit, rows_pad);
    }
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi

extern "C" int mvi_linear_n320_supported(int32_t K, int32_t out_features, int32_t dtype) {
    return out_features == mvi::ln3::kN && K >= 2 * mvi::ln3::kKC && K % mvi::ln3::kKC == 0 && (dtype == MVI_DT_BF16 || dtype == MVI_DT_F16);
}

extern "C" int mvi_linear_n320(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity,
                               int32_t K, int32_t out_features, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                               void* stream) {
    if (rows < 0 || !mvi_linear_n320_supported(K, out_features, dtype))
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: needs out_features = 320, K a multiple of 64 (>= 128), bf16 or f16");
    if (rows == 0) return MVI_OK;
    if (!x || !weight || !out) return mvi::unet_fail(MVI_EINVAL, "linear_n320: NULL pointer");
    if (out_rows_capacity < mvi_ff_geglu_out_rows(rows))
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: out needs room for mvi_ff_geglu_out_rows(rows) rows (whole 256-row blocks are stored)");
    if (x_row_stride < K || out_row_stride < out_features || x_row_stride % 8 || out_row_stride % 8 ||
        ((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out) % 16)
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: x, weight and out rows must be 16-byte aligned");
    if ((int64_t)out_features * K * 2 > 0xFFFFFFFFll || 256 * x_row_stride * 2 > 0xFFFFFFFFll)
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: weight / row block exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == MVI_DT_BF16 ? mvi::linear_n320_launch<__hip_bfloat16>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st)
                                        : mvi::linear_n320_launch<__half>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st);
    return rc ? mvi::unet_fail(rc, "linear_n320: kernel launch failed") : MVI_OK;
}

// Convolutions of token-major activations with C_out a multiple of 320 (all four levels of both networks):
//   * 3x3 / stride 1 / padding 1 (ResBlock in_layers[2] / out_layers[3], svd_inpaint1/sgm/modules/diffusionmodules/openaimodel.py:256-275,
//     :301-318; Upsample.conv :118-134): x [N, H, W, C_in], weight [C_out][9 C_in] = conv.weight.permute(0, 2, 3, 1) flattened;
//   * (3, 1, 1) / padding (1, 0, 0) over frames (the time_stack ResBlock of VideoResBlock, video_model.py:41-54): x [B, T, pixels, C_in],
//     weight [C_out][3 C_in] = conv.weight[:, :, :, 0, 0].permute(0, 2, 1) flattened.
// out [rows, C_out] in rows of out_row_stride elements.
extern "C" int mvi_conv3x3_n320_supported(int32_t C_in, int32_t C_out, int32_t dtype) {
    return C_out > 0 && C_out % mvi::ln3::kN == 0 && C_in >= mvi::ln3::kKC && C_in % mvi::ln3::kKC == 0 &&
           (dtype == MVI_DT_BF16 || dtype == MVI_DT_F16);
}

// K split of a convolution launch: 1 (none) when the (row block, column group) grid already covers half the chip; otherwise enough
// parts to cover it, each with at least 8 chunks of 64.
static int conv_ksplit(int64_t rows, int32_t taps, int32_t C_in, int32_t C_out) {
    static const bool off = getenv("MVI_CONV_KSPLIT") && getenv("MVI_CONV_KSPLIT")[0] == '0';      // same-box A/B runs
    if (off) return 1;
    const int64_t blocks = (rows + mvi::ln3::kRows - 1) / mvi::ln3::kRows * (C_out / mvi::ln3::kN);
    if (blocks >= 128) return 1;
    const int chunks = taps * C_in / mvi::ln3::kKC;
    int ks = (int)(256 / blocks);
    if (ks > 8) ks = 8;
    if (ks > chunks / 8) ks = chunks / 8;
    return ks < 2 ? 1 : ks;
}

static size_t conv_workspace_bytes(int64_t rows, int32_t taps, int32_t C_in, int32_t C_out) {
    const int ks = conv_ksplit(rows, taps, C_in, C_out);
    return ks == 1 ? 0 : (size_t)ks * (size_t)mvi_ff_geglu_out_rows(rows) * (size_t)C_out * sizeof(float);
}

static int conv_taps_n320(const char* what, const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t H, int32_t W,
                          int32_t taps, int32_t stride, int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride,
                          int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    char msg[160];
    auto fail = [&](const char* m) {
        snprintf(msg, sizeof msg, "%s: %s", what, m);
        return mvi::unet_fail(MVI_EINVAL, msg);
    };
    if (N < 0 || H <= 0 || W <= 0 || !mvi_conv3x3_n320_supported(C_in, C_out, dtype) || (stride != 1 && stride != 2))
        return fail("needs C_out a multiple of 320, C_in a multiple of 64, bf16 or f16, stride 1 or 2");
    const int32_t Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;          // (padding 1, kernel 3)
    const int64_t rows = N * Ho * Wo;
    if (rows == 0) return MVI_OK;
    if (!x || !weight || !out) return fail("NULL pointer");
    if (out_rows_capacity < mvi_ff_geglu_out_rows(rows))
        return fail("out needs room for mvi_ff_geglu_out_rows(rows) rows (whole 256-row blocks are stored)");
    if (out_row_stride < C_out || out_row_stride % 8 || ((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out) % 16)
        return fail("x, weight and out rows must be 16-byte aligned");
    if ((int64_t)mvi::ln3::kN * taps * C_in * 2 > 0xFFFFFFFFll || N * H * W * C_in * 2 > 0xFFFFFFFFll || (int64_t)H * W > 0x7FFFFFFFll)
        return fail("weight / activation tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    // the K split is taken when the caller brought its workspace (mvi_conv3x3_n320_workspace_bytes); without one the launch is unsplit
    int ks = conv_ksplit(rows, taps, C_in, C_out);
    if (ks > 1 && (!workspace || workspace_bytes < conv_workspace_bytes(rows, taps, C_in, C_out) || (uintptr_t)workspace % 16)) ks = 1;
    const mvi::ln3::ConvGeom cg = {H, W, C_in / mvi::ln3::kKC, taps, C_out / mvi::ln3::kN, ks, stride, Ho, Wo};
    int rc;
    if (ks > 1)
        rc = dtype == MVI_DT_BF16 ? mvi::linear_n320_launch<__hip_bfloat16, true, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st,
                                                                                        cg, (float*)workspace)
                                  : mvi::linear_n320_launch<__half, true, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st, cg,
                                                                                (float*)workspace);
    else
        rc = dtype == MVI_DT_BF16 ? mvi::linear_n320_launch<__hip_bfloat16, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st, cg)
                                  : mvi::linear_n320_launch<__half, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st, cg);
    if (rc) {
        snprintf(msg, sizeof msg, "%s: kernel launch failed", what);
        return mvi::unet_fail(rc, msg);
    }
    return MVI_OK;
}

extern "C" size_t mvi_conv3x3_n320_workspace_bytes(int64_t N, int32_t H, int32_t W, int32_t C_in, int32_t C_out, int32_t stride) {
    if (N <= 0 || H <= 0 || W <= 0 || C_in <= 0 || C_out <= 0 || (stride != 1 && stride != 2)) return 0;
    return conv_workspace_bytes(N * ((H - 1) / stride + 1) * ((W - 1) / stride + 1), 9, C_in, C_out);
}

extern "C" size_t mvi_conv3t_n320_workspace_bytes(int64_t B, int32_t T, int32_t pixels, int32_t C_in, int32_t C_out) {
    return B <= 0 || T <= 0 || pixels <= 0 || C_in <= 0 || C_out <= 0 ? 0 : conv_workspace_bytes(B * T * pixels, 3, C_in, C_out);
}

extern "C" int mvi_conv3x3_n320(const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t H, int32_t W,
                                int32_t C_in, int32_t C_out, int32_t stride, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                                void* workspace, size_t workspace_bytes, void* stream) {
    return conv_taps_n320("conv3x3_n320", x, weight, bias, out, N, H, W, 9, stride, C_in, C_out, out_rows_capacity, out_row_stride, dtype,
                          workspace, workspace_bytes, stream);
}

extern "C" int mvi_conv3t_n320(const void* x, const void* weight, const float* bias, void* out, int64_t B, int32_t T, int32_t pixels,
                               int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return conv_taps_n320("conv3t_n320", x, weight, bias, out, B, T, pixels, 3, 1, C_in, C_out, out_rows_capacity, out_row_stride, dtype,
                          workspace, workspace_bytes, stream);
}
