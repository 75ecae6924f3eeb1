// Fused GEGLU feed-forward projection for gfx950 (bf16 / f16 MFMA), contraction length 320 — the level-0 model width:
//     out[r, j] = (x[r, :] . W[j, :] + b[j]) * gelu(x[r, :] . W[inner + j, :] + b[inner + j]),   r < rows, j < inner
// Replaces `x, gate = self.proj(x).chunk(2, dim=-1); return x * F.gelu(gate)` of GEGLU
// (svd_inpaint1/sgm/modules/attention.py:87-95) for the 21 level-0 FeedForward layers of the 14 x 576x1024 step
// ([258048, 320] x [320, 2 * 1280]). There the library GEMM is bound by WRITING the [rows, 2 inner] intermediate (1.32 GB per call,
// 0.60 ms = 0.70 PFLOP/s) which geglu_kernel then reads back (0.35 ms): 0.95 ms per call, 20 ms per step. Here the value and the
// gate of an output never leave the accumulators.
//
// Structure (the QK^T half of attn_flash8.hip with x in the role of Q and W in the role of K):
//   * block = 8 waves x 32 rows of x; a wave's 32 rows live in registers as the B operand for the whole block (20 k-steps of 16:
//     80 registers), so x is read from HBM exactly once and never passes through LDS;
//   * W streams through a 3-slot LDS ring by LDS-DMA in tiles of 64 rows x 640 bytes = the 32 value rows and the 32 gate rows of
//     32 outputs (40 KiB; 128-byte groups XOR-swizzled on the SOURCE side: chunk c of row r sits at c ^ ((r >> 1) & 7), conflict-free
//     for ds_read_b128); all 8 waves read the same tile: S^T[w row][x row] = W x^T on v_mfma_f32_32x32x16, 40 MFMAs per wave and step;
//   * two accumulator sets alternate (steps unrolled by two), so the epilogue of step j — bias, erf-GELU, product, pack, 8-byte stores
//     — sits in the same scheduling region as the MFMAs of step j + 1: exact-erf GELU costs about as many issue cycles per output
//     tile as the tile's 40 MFMAs occupy the matrix pipe (K = 320 is short), so it must not run in series with them;
//   * the bias vector (2 inner floats) sits in LDS; one barrier per step; loader waves (4 of 8) issue the tile two steps ahead at the
//     END of a step, behind that step's stores, and wait with a COUNTED vmcnt for everything older than the pieces just issued.
// GELU: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below half an ulp of the bf16 / f16 output by four orders of magnitude)
// with v_rcp_f32 / v_exp_f32: 14 instructions per output, branch-free.
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
namespace ffg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
#define MVI_AS3 __attribute__((address_space(3)))

constexpr int kK = 320;                              // contraction length
constexpr int kKS = kK / 32;                         // MFMA k-steps (v_mfma_f32_16x16x32)
constexpr int kWaves = 8;
constexpr int kRows = 32 * kWaves;                   // x rows per block
constexpr int kStep = 32;                            // outputs per step
constexpr int kRowBytes = kK * 2;                    // 640
constexpr int kTileBytes = 2 * kStep * kRowBytes;    // 40960: 32 value rows | 32 gate rows
constexpr int kPieces = kTileBytes / 1024;           // 40 LDS-DMA pieces of 1 KiB
constexpr int kRing = 3;
constexpr int kLoaders = 4;
constexpr int kPiecesPerLoader = kPieces / kLoaders; // 10

template <typename T> struct Mma;
// c += A B on v_mfma_f32_16x16x32, IN PLACE and in program order (tied inline assembly: csrc/linear_n320.hip, Mma, says why). The compiler
// does not know these are matrix instructions: the wait states around ordinary reads / writes of an accumulator are written out below.
template <> struct Mma<__hip_bfloat16> {
    __device__ static void mfma(f32x4& c, u32x4 a, u32x4 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        bf16x2 r = __builtin_convertvector(f, bf16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
};
template <> struct Mma<__half> {
    __device__ static void mfma(f32x4& c, u32x4 a, u32x4 b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        f16x2 r = __builtin_convertvector(f, f16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
};

__device__ __forceinline__ void dma_piece(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory");
}

// v * gelu(g), gelu(g) = g/2 (1 + erf(g / sqrt 2)); erf(z) = sign(z) (1 - (a1 t + ... + a5 t^5) e^(-z^2)), t = 1 / (1 + p |z|), z = g / sqrt 2
__device__ __forceinline__ float geglu1(float v, float g) {
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(g), 0.3275911f * 0.70710678118654752f, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(g * g * (-0.5f * 1.4426950408889634f));   // e^(-z^2)
    const float erf_abs = __builtin_fmaf(-p, e, 1.0f);
    const float hg = 0.5f * g;                                   // g/2 (1 + sign(g) erf|z|) = g/2 + |g|/2 erf|z|
    return v * __builtin_fmaf(__builtin_fabsf(hg), erf_abs, hg);
}

// kGeglu = true : out[r, j] = (x . W[j] + b[j]) * gelu(x . W[inner + j] + b[inner + j]), j < inner; a step = 32 outputs
// kGeglu = false: out[r, j] = x . W[j] + b[j], j < inner (= the Linear's out_features); a step = 64 outputs (the same two 32-row
//                 blocks of a W tile, both plain) — the bias-only projections of level 0 (packed q/k/v, to_out, proj_in / proj_out),
//                 which the library runs at 0.36 - 0.5 PFLOP/s because at K = 320 they are short loops around a lot of output
// (the timing ablations this kernel was tuned with — every MFMA removed, erf removed, stores removed: DESIGN.md §9 item 2b — were a second copy of
// this file under tools/ff_dev/ until round 6; it is in the git history (commit 93207b4), not kept beside a kernel it would drift from)
template <typename T, bool kGeglu = true>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void ff_geglu_k320_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                          int64_t rows, int inner, int64_t x_rs, int64_t o_rs, int n_blocks) {
    constexpr int kOutStep = kGeglu ? kStep : 2 * kStep;         // outputs (and W rows of the first block) a step advances by
    using M = Mma<T>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    MVI_AS3 char* const lds = (MVI_AS3 char*)smem;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    MVI_AS3 float* const lbias = (MVI_AS3 float*)(lds + kRing * kTileBytes);      // [2 * inner]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;       // the lane's row / column inside a 16 x 16 tile, its 8-element group of a 32-deep k-step
    int bid = blockIdx.x;
    if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // neighbouring row blocks on one XCD (W is shared by all)
    const int64_t row0 = (int64_t)bid * kRows + wave * 32;                       // wave-uniform

    // ---- bias -> LDS (visible after the first barrier of the loop prologue)
    const int n_bias = kGeglu ? 2 * inner : inner;
    for (int i = tid; i < n_bias; i += 64 * kWaves) lbias[i] = bias ? bias[i] : 0.f;

    // ---- x rows: A operand of row tile t, k-step s: element j of lane (n16, kg) = x[row0 + 16 t + n16][32 s + 8 kg + j]
    u32x4 xf[2][kKS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t row = row0 + 16 * t + n16;
        const T* xp = x + (row < rows ? row : rows - 1) * x_rs + 8 * kg;
#pragma unroll
        for (int s = 0; s < kKS; ++s) xf[t][s] = *reinterpret_cast<const u32x4*>(xp + 32 * s);
    }

    // ---- LDS-DMA source addressing: piece p of a tile fills LDS bytes [1024 p, 1024 p + 1024) lane-linearly; the lane's 16 bytes are
    // (tile row r, slot c) and receive source chunk c ^ ((r >> 1) & 7) (low three bits) of the W row tile row r stands for. Round 5
    // (16 x 16 tiles): inside each 32-row block, tile row 16 ct + n is output column 2 n + ct of the step's 32 — column tile 0 holds the
    // even columns and tile 1 the odd ones, so a lane ends up with two ADJACENT outputs of a row and stores them as one dword.
    const bool loader = wave < kLoaders;
    uint32_t p_voff[kPiecesPerLoader];
#pragma unroll
    for (int i = 0; i < kPiecesPerLoader; ++i) {
        const int pc = wave + i * kLoaders;                          // (meaningless for a wave that loads nothing)
        const uint32_t off = 1024u * pc + 16u * lane;
        const uint32_t r = off / kRowBytes, c = (off - r * kRowBytes) >> 4;
        const uint32_t cs = (c & ~7u) | ((c & 7u) ^ ((r >> 1) & 7u));
        const uint32_t col = 2u * (r & 15u) + ((r >> 4) & 1u);       // output column within the block's 32
        const uint32_t wrow = r < 32 ? col : (kGeglu ? (uint32_t)inner + col : 32u + col);
        p_voff[i] = wrow * kRowBytes + 16u * cs;
    }
    const char* const wbase = reinterpret_cast<const char*>(w);
    const int n_steps = inner / kOutStep;
    auto issue_tile = [&](int step) __attribute__((always_inline)) {
        // steps past the end re-load the last tile (never read): every step issues the same number of pieces, the counted wait stays valid
        const int st = step < n_steps ? step : n_steps - 1;
        const uint32_t slot_off = (uint32_t)((step % kRing) * kTileBytes);
        const char* const base = wbase + (int64_t)st * (kOutStep * kRowBytes);
#pragma unroll
        for (int i = 0; i < kPiecesPerLoader; ++i) dma_piece(base, p_voff[i], lds0 + slot_off + 1024u * (wave + i * kLoaders));
    };

    // ---- LDS read addressing: B operand = W rows; lane (n16, kg), block blk, column tile ct, k-step s reads tile row 32 blk + 16 ct + n16,
    // 16-byte chunk 4 s + kg = 8 (s >> 1) + (4 (s & 1) + kg), the low three bits swizzled by (row >> 1) & 7 = n16 >> 1. 640-byte rows
    // alternate between the two halves of the 64 banks like 128-byte ones: the 16 lanes the LDS serves together (four n16 of one kg,
    // eight of the next) land on 16 different 4-bank groups.
    uint32_t ka[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) ka[q] = (uint32_t)(n16 * kRowBytes + (((4 * q + kg) ^ (n16 >> 1)) << 4));
    auto wfrag = [&](uint32_t slot_base, int blk, int ct, int s) __attribute__((always_inline)) {
        return *reinterpret_cast<MVI_AS3 const u32x4*>(lds + slot_base + ka[s & 1] + (s >> 1) * 128 + (32 * blk + 16 * ct) * kRowBytes);
    };

    // Output addressing. A operand = x rows, B operand = W rows: the OUTPUT COLUMN sits on the lane — columns 2 n16 and 2 n16 + 1 of the
    // step's 32 (column tiles 0 and 1) — and accumulator register r of row tile t is x row 16 t + 4 kg + r of the wave's 32: one bias
    // pair per lane and block, and a store instruction writes four rows x 16 dwords (64 bytes each). Stores are not predicated: the
    // caller provides an output with room for the rows of whole blocks (include/mvi_unet_ops.h).
    char* const obase = reinterpret_cast<char*>(out + row0 * o_rs);
    const int64_t orow_bytes = o_rs * 2;
    const uint32_t lane_off = (uint32_t)(4 * kg * orow_bytes + 4 * n16);         // + (16 t + r) rows

    // One step = the 80 MFMAs of this step's outputs (first block and second block — value and gate — x two column tiles x two row tiles:
    // eight accumulator chains; every W fragment feeds the two row tiles; fragments requested kAhead reads ahead) with the EPILOGUE OF
    // THE PREVIOUS STEP cut into 8 slices between them: k-step s carries rows (t, r) = (s / 4, s % 4) of the previous step (bias is
    // already in the accumulator: its chain starts from it). Scheduling fences keep the slices where they are — left alone, the
    // scheduler issues the MFMAs back to back and the ~300 VALU instructions of the epilogue behind them, and the step takes the sum
    // of both (measured on the 32x32 form: 637 us per call).
    constexpr int kAhead = 3;
    // Plain form: a step is 64 output columns = one full 128-byte line per row. The tiles go through a wave-private 4 KiB LDS tile
    // [32 rows][64 columns] as dwords (two adjacent columns) and leave as four 16-byte stores per lane, each instruction covering
    // 8 rows x 128 contiguous bytes. A dword store's four kg sit 4 rows = 512 bytes apart, in the same banks: rows with odd kg keep
    // their two 64-byte halves swapped, and the reader undoes it.
    const uint32_t otile = (uint32_t)(kRing * kTileBytes + n_bias * (int)sizeof(float) + wave * 4096);
    const uint32_t ot_w = otile + (uint32_t)(4 * kg * 128 + 4 * n16);              // + 128 (16 t + r); first block at 0, second at 64, ^ 64 (kg & 1)
    const uint32_t ot_r = otile + (uint32_t)((lane >> 3) * 128 + (((lane & 7) ^ (4 * ((lane >> 5) & 1))) << 4));   // + 1024 i: rows 8 i + lane / 8
    const uint32_t st_off = (uint32_t)((lane >> 3) * orow_bytes + (lane & 7) * 16);
    auto put_dword = [&](int blk, int t, int r, uint32_t pk) __attribute__((always_inline)) {
        *reinterpret_cast<MVI_AS3 uint32_t*>(lds + ot_w + 128 * (16 * t + r) + ((64 * blk) ^ (64 * (kg & 1)))) = pk;
    };
    auto flush_tile = [&](char* op) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ot_r + 1024 * i);
            *reinterpret_cast<u32x4*>(op + (8 * i) * orow_bytes + st_off) = v;
        }
    };
    // rows (t, r) of the previous step: GEGLU — gate the two adjacent columns and store them as one dword; plain — both blocks to the tile
    auto out_rows = [&](char* op, int t, int r, const f32x4 (&pv)[2][2], const f32x4 (&pg)[2][2]) __attribute__((always_inline)) {
        if (kGeglu) {
            const uint32_t pk = M::pack2(geglu1(pv[0][t][r], pg[0][t][r]), geglu1(pv[1][t][r], pg[1][t][r]));
            *reinterpret_cast<uint32_t*>(op + (16 * t + r) * orow_bytes + lane_off) = pk;
        } else {
            put_dword(0, t, r, M::pack2(pv[0][t][r], pv[1][t][r]));
            put_dword(1, t, r, M::pack2(pg[0][t][r], pg[1][t][r]));
        }
    };
    auto step_fn = [&](auto with_prev_c, uint32_t slot_base, int step, f32x4 (&av)[2][2], f32x4 (&ag)[2][2], const f32x4 (&pv)[2][2],
                       const f32x4 (&pg)[2][2]) __attribute__((always_inline)) {
        constexpr bool kPrev = decltype(with_prev_c)::value;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int n = step * kOutStep + 2 * n16 + ct;
            const float bv = lbias[n], bg = lbias[(kGeglu ? inner : kStep) + n];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) { av[ct][t][i] = bv; ag[ct][t][i] = bg; }
        }
        // (vector write -> matrix read of the accumulators, by hand)
        asm volatile("s_nop 4" : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[1][0]), "+v"(av[1][1]), "+v"(ag[0][0]), "+v"(ag[0][1]), "+v"(ag[1][0]),
                     "+v"(ag[1][1]));
        char* const op = obase + (step - 1) * (kOutStep * 2);        // the previous step's output columns
        // fragment q = 4 s + 2 blk + ct
        u32x4 wf[kAhead + 1];
#pragma unroll
        for (int q = 0; q < kAhead; ++q) wf[q] = wfrag(slot_base, (q >> 1) & 1, q & 1, q >> 2);
#pragma unroll
        for (int s = 0; s < kKS; ++s) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int q = 4 * s + h, blk = h >> 1, ct = h & 1;
                if (q + kAhead < 4 * kKS) wf[(q + kAhead) % (kAhead + 1)] = wfrag(slot_base, ((q + kAhead) >> 1) & 1, (q + kAhead) & 1, (q + kAhead) >> 2);
                f32x4 (&a)[2][2] = blk ? ag : av;
                M::mfma(a[ct][0], xf[0][s], wf[q % (kAhead + 1)]);
                M::mfma(a[ct][1], xf[1][s], wf[q % (kAhead + 1)]);
            }
            if (kPrev && s < 8) out_rows(op, s >> 2, s & 3, pv, pg);
            if (kPrev && !kGeglu && s == 8) flush_tile(op);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the last step's outputs, with nothing left to hide behind
    auto drain = [&](int step, const f32x4 (&pv)[2][2], const f32x4 (&pg)[2][2]) __attribute__((always_inline)) {
        char* const op = obase + step * (kOutStep * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u) out_rows(op, u >> 2, u & 3, pv, pg);
        if (!kGeglu) flush_tile(op);
    };
    auto close_step = [&](int step) __attribute__((always_inline)) {
        // loaders: issue the tile two steps ahead (its slot held step - 1, which nobody reads any more), then wait for everything
        // older than those pieces — this wave's pieces of tile step + 1 among them; then the block meets. (The s_nops: the step's last
        // matrix instructions have written their accumulators before the next step's epilogue slices read them.)
        if (loader) {
            issue_tile(step + 2);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_nop 7\n\ts_nop 7\n\ts_barrier" ::"n"(kPiecesPerLoader) : "memory");
        } else {
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);                           // (nothing of the next step's epilogue slices moves above the wait states)
    };

    // ---- prologue: tiles 0 and 1 in flight, tile 0 landed
    if (loader) {
        issue_tile(0);
        issue_tile(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesPerLoader) : "memory");
    }
    __syncthreads();                                             // tile 0 and the bias vector are in LDS

    f32x4 va[2][2], ga[2][2], vb[2][2], gb[2][2];                // accumulator sets A (even steps) and B (odd steps): [column tile][row tile]
    auto next_slot = [&](uint32_t s) __attribute__((always_inline)) { return s + kTileBytes == (uint32_t)(kRing * kTileBytes) ? 0u : s + kTileBytes; };
    auto settle = [&](f32x4 (&v)[2][2], f32x4 (&g)[2][2]) __attribute__((always_inline)) {
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[1][0]),
                     "+v"(g[1][1]));
    };
    int j = 0;
    uint32_t slot = 0;                                           // byte offset of step j's ring slot
    step_fn(std::false_type{}, slot, 0, va, ga, va, ga);
    close_step(0);
    slot = next_slot(slot);
    for (j = 1; j + 1 < n_steps; j += 2) {
        step_fn(std::true_type{}, slot, j, vb, gb, va, ga);
        close_step(j);
        slot = next_slot(slot);
        step_fn(std::true_type{}, slot, j + 1, va, ga, vb, gb);
        close_step(j + 1);
        slot = next_slot(slot);
    }
    if (j < n_steps) {                                           // one step left (n_steps even): it is an odd step
        step_fn(std::true_type{}, slot, j, vb, gb, va, ga);
        settle(vb, gb);
        drain(j, vb, gb);
    } else {
        settle(va, ga);
        drain(j - 1, va, ga);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // trailing (unused) pieces land before the block releases its LDS
}

}  // namespace ffg

template <typename T, bool kGeglu>
static int ff_k320_launch(const void* x, const void* w, const float* bias, void* out, int64_t rows, int inner, int64_t x_rs,
                          int64_t o_rs, hipStream_t st) {
    using namespace ffg;
    const int64_t n_blocks = (rows + kRows - 1) / kRows;
    if (n_blocks > 0x7FFFFFFFll) return MVI_EINVAL;
    const int lds_bytes = kRing * kTileBytes + (kGeglu ? 2 : 1) * inner * (int)sizeof(float) + (kGeglu ? 0 : kWaves * 4096);
    static unsigned long long attr_set = 0;                      // per device and instantiation: the opt-in for > 64 KiB of dynamic LDS
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MVI_EHIP;
    auto kern = &ff_geglu_k320_kernel<T, kGeglu>;
    if (!(attr_set >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MVI_EHIP;
        attr_set |= 1ull << dev;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks), dim3(64 * kWaves), lds_bytes, st, (const T*)x, (const T*)w, bias, (T*)out, rows, inner,
                       x_rs, o_rs, (int)n_blocks);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi

extern "C" int mvi_ff_geglu_supported(int32_t K, int32_t inner, int32_t dtype) {
    return K == mvi::ffg::kK && inner > 0 && inner % mvi::ffg::kStep == 0 && inner >= 2 * mvi::ffg::kStep &&
           mvi::ffg::kRing * mvi::ffg::kTileBytes + 2 * (int64_t)inner * 4 <= 160 * 1024 && (dtype == MVI_DT_BF16 || dtype == MVI_DT_F16);
}

extern "C" int64_t mvi_ff_geglu_out_rows(int64_t rows) { return (rows + mvi::ffg::kRows - 1) / mvi::ffg::kRows * mvi::ffg::kRows; }

extern "C" int mvi_ff_geglu(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity,
                            int32_t K, int32_t inner, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, void* stream) {
    if (rows < 0 || !mvi_ff_geglu_supported(K, inner, dtype))
        return mvi::unet_fail(MVI_EINVAL, "ff_geglu: needs K = 320, inner a multiple of 32 (64 ... 4864), bf16 or f16");
    if (rows == 0) return MVI_OK;
    if (!x || !weight || !out) return mvi::unet_fail(MVI_EINVAL, "ff_geglu: NULL pointer");
    if (out_rows_capacity < mvi_ff_geglu_out_rows(rows))
        return mvi::unet_fail(MVI_EINVAL, "ff_geglu: out needs room for mvi_ff_geglu_out_rows(rows) rows (whole 256-row blocks are stored)");
    if (x_row_stride < K || out_row_stride < inner || x_row_stride % 8 || out_row_stride % 4 ||
        ((uintptr_t)x | (uintptr_t)weight) % 16 || (uintptr_t)out % 8)
        return mvi::unet_fail(MVI_EINVAL, "ff_geglu: rows must be 16-byte aligned (x, weight) / 8-byte aligned (out)");
    if ((int64_t)2 * inner * K * 2 > 0xFFFFFFFFll) return mvi::unet_fail(MVI_EINVAL, "ff_geglu: weight exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == MVI_DT_BF16
                       ? mvi::ff_k320_launch<__hip_bfloat16, true>(x, weight, bias, out, rows, inner, x_row_stride, out_row_stride, st)
                       : mvi::ff_k320_launch<__half, true>(x, weight, bias, out, rows, inner, x_row_stride, out_row_stride, st);
    return rc ? mvi::unet_fail(rc, "ff_geglu: kernel launch failed") : MVI_OK;
}

extern "C" int mvi_linear_k320_supported(int32_t K, int32_t out_features, int32_t dtype) {
    return K == mvi::ffg::kK && out_features >= 4 * mvi::ffg::kStep && out_features % (2 * mvi::ffg::kStep) == 0 &&
           mvi::ffg::kRing * mvi::ffg::kTileBytes + (int64_t)out_features * 4 + mvi::ffg::kWaves * 4096 <= 160 * 1024 &&
           (dtype == MVI_DT_BF16 || dtype == MVI_DT_F16);
}

extern "C" int mvi_linear_k320(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity,
                               int32_t K, int32_t out_features, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                               void* stream) {
    if (rows < 0 || !mvi_linear_k320_supported(K, out_features, dtype))
        return mvi::unet_fail(MVI_EINVAL, "linear_k320: needs K = 320, out_features a multiple of 64 (128 ... 2048), bf16 or f16");
    if (rows == 0) return MVI_OK;
    if (!x || !weight || !out) return mvi::unet_fail(MVI_EINVAL, "linear_k320: NULL pointer");
    if (out_rows_capacity < mvi_ff_geglu_out_rows(rows))
        return mvi::unet_fail(MVI_EINVAL, "linear_k320: out needs room for mvi_ff_geglu_out_rows(rows) rows (whole 256-row blocks are stored)");
    if (x_row_stride < K || out_row_stride < out_features || x_row_stride % 8 || out_row_stride % 4 ||
        ((uintptr_t)x | (uintptr_t)weight) % 16 || (uintptr_t)out % 8)
        return mvi::unet_fail(MVI_EINVAL, "linear_k320: rows must be 16-byte aligned (x, weight) / 8-byte aligned (out)");
    if ((int64_t)out_features * K * 2 > 0xFFFFFFFFll) return mvi::unet_fail(MVI_EINVAL, "linear_k320: weight exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == MVI_DT_BF16
                       ? mvi::ff_k320_launch<__hip_bfloat16, false>(x, weight, bias, out, rows, out_features, x_row_stride, out_row_stride, st)
                       : mvi::ff_k320_launch<__half, false>(x, weight, bias, out, rows, out_features, x_row_stride, out_row_stride, st);
    return rc ? mvi::unet_fail(rc, "linear_k320: kernel launch failed") : MVI_OK;
}
