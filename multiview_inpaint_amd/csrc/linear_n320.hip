// Linear with 320 outputs and a long contraction (K = 1280: the second projection of the 21 level-0 FeedForward layers and their
// temporal twins, svd_inpaint1/sgm/modules/attention.py:98-115 `net[2]`) for gfx950, bf16 / f16 MFMA:
//     out[r, n] = x[r, :] . W[n, :] + b[n],   r < rows, n < 320, K a multiple of 64
// Why it exists: at [258048, 1280] x [1280, 320] the library's 256 x 256 macro-tile covers 320 columns with two column tiles, 37 % of
// the second one empty, and runs at 0.60 PFLOP/s (0.35 ms per call, 42 calls = 14.9 ms of a 168 ms step). Here a block's 256 rows
// keep ALL 320 outputs in accumulators (a wave: 2 row tiles x 20 column tiles of 16 x 16 = 160 registers), so nothing is padded and x
// is read once.
//
// Structure (csrc/ff_geglu.hip with the roles turned: there x is stationary and the outputs stream, here the outputs are stationary
// and K streams):
//   * block = 8 waves x 32 rows; K advances in chunks of 64; per chunk and wave 2 k-steps x 20 column tiles x 2 row tiles = 80 MFMAs
//     (v_mfma_f32_16x16x32, A = x rows, B = W rows: the output column sits on the lane; every W fragment read from LDS feeds two of
//     them). Round 5: 16x16x32 instead of 32x32x16 — the same FLOPs per cycle, LDS reads and registers, but the chip holds a higher
//     clock on this shape under load (MI355X_MICROARCH.md, DVFS item 7): every form of this kernel 9 - 11 % faster on the same box
//     (profiles/round5_n320_mfma16_ab.txt). The MFMAs are tied inline assembly (see Mma);
//   * W chunk [320 rows][64 k] = 40 KiB streams through a 3-slot LDS ring by LDS-DMA (128-byte rows, 16-byte chunk c of row r at
//     slot c ^ ((r >> 1) & 7), swizzled on the source side: the K image of attn_flash8.hip, conflict-free ds_read_b128); all 8 waves
//     read the same chunk; 4 loader waves issue the chunk two ahead at the END of a chunk and wait with a counted vmcnt;
//   * a wave's own x rows go HBM -> registers directly (4 x global_load_dwordx4 per chunk — 2 row tiles x 2 k-steps, a lane holds
//     rows n16 and 16 + n16 — one chunk ahead, double-buffered; the 4 loads of a chunk touch the same 32 lines). They are inline assembly like the DMA: a load the compiler knows about
//     makes it wait for vmcnt(0) at the first use — the DMA pieces just issued included;
//   * the bias is the accumulators' initial value; outputs leave through a wave-private 4 KiB LDS tile as 16-byte stores
//     (four column tiles = one 128-byte line per row per flush) into a buffer padded to whole 256-row blocks.
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


typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define MVI_AS3 __attribute__((address_space(3)))

constexpr int kN = 320;                              // outputs
constexpr int kNT = kN / 16;                         // column tiles (16 wide) per wave
constexpr int kWaves = 8;
constexpr int kRows = 32 * kWaves;                   // x rows per block
constexpr int kKC = 64;                              // contraction elements per chunk
constexpr int kChunkBytes = kN * kKC * 2;            // 40960
constexpr int kPieces = kChunkBytes / 1024;          // 40 LDS-DMA pieces of 1 KiB (8 W rows each)
constexpr int kRing = 3;
constexpr int kLoaders = 4;
constexpr int kPiecesPerLoader = kPieces / kLoaders; // 10
constexpr int kLdsBytes = kRing * kChunkBytes + kWaves * 4096;

template <typename T> struct Mma;
template <> struct Mma<__hip_bfloat16> {
    using frag = bf16x8;
    // c += A B, IN PLACE and in program order: through the builtin the register allocator took the untied form for two MFMAs in three,
    // rotated the forty 4-register accumulators through the W fragments' registers (write-after-read stalls on the next ds_read) and
    // spilled; as volatile assembly the loop below is issued as written. The compiler does not know these are matrix instructions:
    // the wait states between the last of them and the first ordinary read of an accumulator are in the kernel (mfma_settle).
    __device__ static void mfma(f32x4& c, u32x4 a, u32x4 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
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
    __device__ static void mfma(f32x4& c, u32x4 a, u32x4 b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        f16x2 r = __builtin_convertvector(f, f16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
    __device__ static float lo(uint32_t w) { f16x2 h = *reinterpret_cast<f16x2*>(&w); return (float)h[0]; }
    __device__ static float hi(uint32_t w) { f16x2 h = *reinterpret_cast<f16x2*>(&w); return (float)h[1]; }
};

// a pointer every lane of the wave holds the same value of, said to the compiler (the asynchronous loads take their base in scalar registers)
__device__ __forceinline__ const char* wave_uniform(const char* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void dma_piece(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(wave_uniform((const char*)sbase)), "v"(voff), "s"(lds_addr) : "memory");
}
// 16 bytes per lane, invisible to the compiler's wait-count bookkeeping (see the header): whoever reads the result waits first
__device__ __forceinline__ u32x4 load16_async(const void* sbase, uint32_t voff) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r) : "v"(voff), "s"(wave_uniform((const char*)sbase)) : "memory");
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
    int cc_major;                                    // K ordered (channel chunk, tap, 64 channels) instead of (tap, channel): see conv_k_order
    int frame_major;                                 // taps == 3: row blocks walked (pixel block, frame) instead of (frame, pixel block)
    // Round 6, split-operand convolutions at fp32 accuracy (the first-stage decoder, svd/vae_split.py):
    int dup3;                                        // the LOGICAL channel axis is (hi | hi | lo) over a PHYSICAL x row of (hi | lo): logical 64-channel
                                                     // chunk cc of a tap reads physical chunk cc - cpc / 3 once cc >= cpc / 3 (cpc = 3 C / 64)
    int out_cols;                                    // > 0 (with kSplit): the fp32 accumulators ARE the result — stored to `part` [ksplit][padded
                                                     // rows][out_cols], columns >= out_cols (the padding of C_out to whole column groups) dropped, no reduction
};
// kUps (round 6, the token stream's Upsample.conv: openaimodel.py:107-150): cg.H, cg.W are those of the nearest-neighbour 2x UPSAMPLED image, x
// holds the (H / 2) x (W / 2) source — pixel (y, x) reads source (y >> 1, x >> 1): the convolution of F.interpolate(x, scale_factor=2)
// without the 4x tensor ever existing. An instantiation of its own (3 launches per step): the other convolutions carry none of its registers.

// kStats: the GroupNorm that FOLLOWS this convolution gets its statistics from here (ResBlock out_layers[0] behind in_layers[2],
// openaimodel.py:292-305, :339-343; the temporal twin, video_model.py:41-54): every block leaves, per GroupNorm group inside its 320
// output channels, (count, mean, M2) of its 256 rows x C_g channels of the ROUNDED outputs (what the norm will read) + the norm's
// per-(sample, channel) bias (the timestep embedding rides inside the norm), in the layout gt_merge_kernel of
// csrc/groupnorm_tokens.hip takes: part[(sample * chunks + chunk) * G + group][3], chunk = the block's place among the S / 256 blocks
// of its sample (the host only takes this form when S % 256 == 0). The separate statistics pass over the tensor disappears.
struct GnStats {
    float* part;                                     // [samples * (S / 256) * G][3]; NULL: no statistics
    const float* chan_bias;                          // [samples, C_out] or NULL
    int G, S;                                        // GroupNorm groups over all C_out channels; rows (pixels) per sample
};

// kGeglu (round 5): the GEGLU projection of the level-1 / level-2 FeedForward layers (K = 640 / 1280; attention.py:87-95,
// `x, gate = self.proj(x).chunk(2, dim=-1); x * F.gelu(gate)`) in this kernel's frame: a block's 320 accumulator columns are 160
// value columns and the 160 gate columns of the SAME outputs (W rows 160 g .. and inner + 160 g ..: five column tiles each, value
// tile t and gate tile t + 5 meet in the same lane and register), the epilogue gates in registers and stores 160 outputs per row.
// The [rows, 2 inner] intermediate the library GEMM writes and geglu_kernel reads back (1 GB per call at level 1) never exists.
// cg.groups = inner / 160 column groups; W is [2 inner][K], bias [2 inner] or NULL.
// v * gelu(g), exact-erf GELU by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): the arithmetic of csrc/ff_geglu.hip's geglu1
__device__ __forceinline__ float geglu1(float v, float g) {
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(g), 0.3275911f * 0.70710678118654752f, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(g * g * (-0.5f * 1.4426950408889634f));
    const float erf_abs = __builtin_fmaf(-p, e, 1.0f);
    const float hg = 0.5f * g;
    return v * __builtin_fmaf(__builtin_fabsf(hg), erf_abs, hg);
}

// kLn (round 5): the projection's output never reaches memory — the residual add(s) and the LayerNorm that FOLLOW it in the transformer
// blocks (attention.py:544-572 `x = attn1(norm1(x)) + x; ... norm3(x)`, video_attention.py:110-141) run in the epilogue, on the
// arithmetic of csrc/token_rows.hip's add_layernorm (which this replaces at level 0, where a block holds whole rows of 320):
//     h = round(x W^T + bias);  s_pre = round(resid + h);  s = round(s_pre + row[r / row_div]);  y = LayerNorm(s) * ln_w + ln_b
// resid / row optional (without resid: s_pre = h); s_pre / s are stored when asked for, y always (`out`). All of resid, s_pre, s, y
// are [rows, 320] in rows of o_rs elements (the outputs padded to whole blocks like `out`).
struct LnEpi {
    const void *resid, *row;                         // [rows, 320] or NULL; [G, 320] or NULL (row r takes row[r / row_div])
    int64_t row_div;
    const float *w, *b;                              // LayerNorm weight / bias [320]
    void *s_pre, *s;                                 // optional outputs
    float eps;
};

// NOUT (round 6): the block's column count — 320 everywhere in the UNet; 256 for the first-stage decoder's 128 / 256 / 512-channel
// convolutions (split operands, fp32 out). Everything below is written in terms of kN / kNT / kChunkBytes / kPieces..., re-derived here
// from NOUT (the epilogues that know about 320 — kStats, kGeglu, kLn — are only instantiated with it).
// The same arithmetic on a PAIR of outputs with the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue
// slot). Round 5 found them an anti-lever BESIDE MFMAs (csrc/ff_geglu.hip's interleaved epilogue); this kernel's GEGLU epilogue runs after
// the last MFMA of the block, with nothing on the matrix pipe. The two values of a pair are adjacent accumulator registers (rows r, r + 1
// of one tile), so no register moves are needed. |v| enters through the scalar fma's abs modifier (the packed forms have none).
typedef float f32p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32p geglu2(f32p v, f32p g) {
    const f32p d = {__builtin_fmaf(__builtin_fabsf(g.x), 0.3275911f * 0.70710678118654752f, 1.0f),
                    __builtin_fmaf(__builtin_fabsf(g.y), 0.3275911f * 0.70710678118654752f, 1.0f)};
    const f32p t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32p p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    p = p * t;
    const f32p gg = g * g * (-0.5f * 1.4426950408889634f);
    const f32p e = {__builtin_amdgcn_exp2f(gg.x), __builtin_amdgcn_exp2f(gg.y)};
    const f32p erf_abs = 1.0f - p * e;
    const f32p hg = 0.5f * g;
    const f32p s = {__builtin_fmaf(__builtin_fabsf(hg.x), erf_abs.x, hg.x), __builtin_fmaf(__builtin_fabsf(hg.y), erf_abs.y, hg.y)};
    return v * s;
}
static const int g_geglu_packed = [] { const char* e = getenv("MVI_GEGLU_PACKED"); return (e && e[0] == '0') ? 0 : 1; }();

template <typename T, bool kConv, bool kSplit = false, bool kStats = false, bool kGeglu = false, bool kLn = false, int NOUT = 320,
          bool kUps = false, bool kPersist = false>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void linear_n320_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                        int64_t rows, int K, int64_t x_rs, int64_t o_rs, int n_blocks, ConvGeom cg, float* __restrict__ part, GnStats gn,
                        LnEpi ln = {nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, 0.f}) {
    using M = Mma<T>;
    static_assert(NOUT == 320 || !(kStats || kGeglu || kLn), "the fused epilogues are written for 320 columns");
    static_assert(!kPersist || !(kConv || kSplit || kStats || kLn), "the persistent form is written for the plain and the GEGLU projection");
    constexpr int kN = NOUT, kNT = kN / 16, kChunkBytes = kN * kKC * 2, kPieces = kChunkBytes / 1024, kPiecesPerLoader = kPieces / kLoaders;
    static_assert(kNT % 4 == 0 && kPieces % kLoaders == 0, "column tiles leave four at a time; every loader moves the same number of pieces");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    MVI_AS3 char* const lds = (MVI_AS3 char*)smem;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;       // the lane's row / column inside a 16 x 16 tile, its 8-element group of a 32-deep k-step
    int bid = blockIdx.x;
    // kPersist (round 6): the grid is one block per CU (a multiple of 8) and a block walks tiles — tile = what a block of the plain grid
    // computes — of its XCD's contiguous share of them: lo + l, lo + l + per, ... (l = the block's place among its XCD's blocks), so the
    // blocks that run together on an XCD are still neighbouring row blocks. While a tile's last two chunks are computed the NEXT tile's
    // first two W chunks and first x rows are requested in place of the re-loads of the last chunk the plain form issues there (same
    // number of requests, same counted waits): the next tile starts behind the epilogue without a prologue.
    int tile_hi = 0, tile_per = 0;
    if (kPersist) {
        tile_per = (int)(gridDim.x >> 3);
        const int xcd = (int)(blockIdx.x & 7);
        bid = (int)((int64_t)xcd * n_blocks / 8) + (int)(blockIdx.x >> 3);
        tile_hi = (int)((int64_t)(xcd + 1) * n_blocks / 8);
    } else if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // neighbouring row blocks on one XCD (W is shared by all)
    int tile = bid;                                  // (kPersist: the tile in hand)
    const T* const w_all = w;
    T* const out_all = out;
    const float* const bias_all = bias;
    int part_col0 = 0, ks = 0;
    int gg = 0;                                      // kGeglu: the block's column group (160 outputs)
    constexpr int kHalf = kN / 2;                    // 160
    const int inner = kGeglu ? cg.groups * kHalf : 0;
    if (kGeglu) {
        gg = bid % cg.groups;
        bid /= cg.groups;
        out += gg * kHalf;
    }
    if (!kGeglu && cg.groups > 1) {                  // (row block, column group): the groups of one row block run side by side
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
    if (kConv && cg.taps == 3 && cg.frame_major) {
        // (3,1,1) form: the three taps of a block read the SAME 256 pixels of frames t - 1, t, t + 1, i.e. every activation row is read by
        // the blocks of three consecutive frames. In row order those blocks are S / 256 launches apart and each finds the rows evicted
        // from L2; walked (pixel block, frame) instead, they are neighbours — dispatched back to back, on one XCD (the remap above) —
        // and two of the three reads hit. Only the order of the row blocks changes (the host allows it when S % 256 == 0).
        const int bpf = cg.W / kRows, per_video = cg.H * bpf;                    // cg.H frames of cg.W pixels
        const int v = bid / per_video, jj = bid - v * per_video;
        const int pb = jj / cg.H, f = jj - pb * cg.H;
        bid = v * per_video + f * bpf + pb;
    }
    int64_t row0 = (int64_t)bid * kRows + wave * 32;                             // wave-uniform
    // this block's chunks: c0 .. c0 + n_chunks - 1 of the K / 64 (all of them unless kSplit); chunk indices below are relative to c0
    const int c0 = kSplit ? (int)((int64_t)ks * (K / kKC) / cg.ksplit) : 0;
    const int n_chunks = kSplit ? (int)((int64_t)(ks + 1) * (K / kKC) / cg.ksplit) - c0 : K / kKC;

    // ---- accumulators start from the bias: tile (t, j) = rows 16 t .., columns 16 j ..; the lane holds column 16 j + n16 of rows
    // 16 t + 4 kg + r (r = register 0 .. 3)
    f32x4 acc[2][kNT];
#pragma unroll
    for (int j = 0; j < kNT; ++j) {
        const float b = bias && !kSplit ? (kGeglu ? bias[(j < kNT / 2 ? 0 : inner - kHalf) + gg * kHalf + 16 * j + n16] : bias[16 * j + n16])
                                        : 0.f;                           // (kSplit: the reduction adds it)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][j][r] = b;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the compiler's own loads are done before the hand-counted ones start
    // kPersist: the bias of the NEXT tile waits in LDS (two 1 KiB pieces behind the waves' output tiles: plain — columns 0 .. 255 | 256 ..
    // 319; kGeglu — the 160 value columns | the 160 gate columns), brought by wave 0 with the DMA during the tile before: a global load
    // at the start of a tile would have to wait for the epilogue's stores (one counter for both)
    constexpr uint32_t kBiasLds = (uint32_t)(kRing * kChunkBytes + kWaves * 4096);

    // ---- x: A operand of row tile t, k-step s (32 deep) of chunk c: element e of lane (n16, kg) = x[row0 + 16 t + n16][64 c + 32 s + 8 kg + e];
    // rows past the end read the last row (kConv: the base is the tensor and the lane offset absolute — a tap's row may lie before the
    // wave's first one)
    const char* xbase = kConv ? reinterpret_cast<const char*>(x)
                              : reinterpret_cast<const char*>(x + (row0 < rows ? row0 : rows - 1) * x_rs);   // wave-uniform
    uint32_t x_voff[2];
    uint32_t tap_ok[2] = {0x1FFu, 0x1FFu};           // bit 3 (dy + 1) + (dx + 1): that neighbour of the row's pixel is inside the image
    uint32_t ups_par = 0;                            // kUps: bits 2 t + 1, 2 t = parity of row tile t's pixel (y, x) in the upsampled image
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t row = row0 + 16 * t + n16;
        const int64_t rclamp = kConv ? (row < rows ? row : rows - 1)
                                     : (row < rows ? (row - (row0 < rows ? row0 : rows - 1)) : 0);         // (a block never starts past the end)
        x_voff[t] = (uint32_t)(rclamp * x_rs * 2 + 16 * kg);
        if (kConv) {
            const int64_t img = rclamp / ((int64_t)cg.Ho * cg.Wo);
            const int pix = (int)(rclamp - img * ((int64_t)cg.Ho * cg.Wo));
            const int py = pix / cg.Wo * cg.stride, px = (pix - pix / cg.Wo * cg.Wo) * cg.stride;  // the centre, in the input image
            if (cg.stride != 1) x_voff[t] = (uint32_t)(((img * cg.H + py) * cg.W + px) * x_rs * 2 + 16 * kg);
            if (kUps) {
                x_voff[t] = (uint32_t)(((img * (cg.H >> 1) + (py >> 1)) * (cg.W >> 1) + (px >> 1)) * x_rs * 2 + 16 * kg);
                ups_par |= (uint32_t)(((py & 1) << 1) | (px & 1)) << (2 * t);
            }
            if (cg.taps == 9) {
                if (py == 0) tap_ok[t] &= ~0x007u;
                if (py == cg.H - 1) tap_ok[t] &= ~0x1C0u;
                if (px == 0) tap_ok[t] &= ~0x049u;
                if (px == cg.W - 1) tap_ok[t] &= ~0x124u;
            } else {                                 // bit dy + 1
                if (py == 0) tap_ok[t] &= ~0x1u;
                if (py == cg.H - 1) tap_ok[t] &= ~0x4u;
            }
        }
    }
    // kConv: (tap, channel chunk) of the next load_x call — they come in chunk order: chunk c = tap * cpc + cc, or cc * taps + tap (cc_major)
    int ld_tap = kConv ? (cg.cc_major ? c0 % cg.taps : c0 / cg.cpc) : 0, ld_cc = kConv ? (cg.cc_major ? c0 / cg.taps : c0 - ld_tap * cg.cpc) : 0;
    // xr[2 t + s]; keep[t] = the lane's keep mask for row tile t's fragments (all ones unless kConv and the tap is outside the image)
    auto load_x = [&](int c, u32x4 (&xr)[4], uint32_t (&keep)[2]) __attribute__((always_inline)) {
        if (!kConv) {
            const char* const base = xbase + (int64_t)(c0 + c) * (kKC * 2);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) xr[2 * t + s] = load16_async(base + 64 * s, x_voff[t]);
            keep[0] = keep[1] = ~0u;
            return;
        }
        // (both counters are wave-uniform by construction; said explicitly: the base below is a scalar operand of the loads)
        const int tap_u = __builtin_amdgcn_readfirstlane(ld_tap), cc_u = __builtin_amdgcn_readfirstlane(ld_cc);
        const int dy = cg.taps == 9 ? tap_u / 3 - 1 : tap_u - 1, dx = cg.taps == 9 ? tap_u - 3 * (tap_u / 3) - 1 : 0;
        const int delta = (dy * cg.W + dx) * (int)(x_rs * 2);
        const int third = cg.cpc / 3;
        const int cc_phys = cg.dup3 && cc_u >= third ? cc_u - third : cc_u;       // (hi | hi | lo) read from (hi | lo)
        const char* const base = xbase + cc_phys * (kKC * 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool ok = (tap_ok[t] >> tap_u) & 1u;
            int delta_t = delta;
            if (kUps) {
                // neighbour (y + dy, x + dx) of the upsampled image lives at source ((y + dy) >> 1, (x + dx) >> 1): one source row / pixel
                // back only from an even y / x, one forward only from an odd one
                const int py1 = (int)((ups_par >> (2 * t + 1)) & 1u), px1 = (int)((ups_par >> (2 * t)) & 1u);
                const int ddy = dy < 0 ? py1 - 1 : (dy > 0 ? py1 : 0), ddx = dx < 0 ? px1 - 1 : (dx > 0 ? px1 : 0);
                delta_t = (ddy * (cg.W >> 1) + ddx) * (int)(x_rs * 2);
            }
            const uint32_t voff = ok ? x_voff[t] + (uint32_t)delta_t : x_voff[t];
#pragma unroll
            for (int s = 0; s < 2; ++s) xr[2 * t + s] = load16_async(base + 64 * s, voff);
            keep[t] = ok ? ~0u : 0u;
        }
        if (c + 1 < n_chunks) {                      // (the calls past the end repeat the last chunk)
            if (cg.cc_major) { if (++ld_tap == cg.taps) { ld_tap = 0; ++ld_cc; } }
            else if (++ld_cc == cg.cpc) { ld_cc = 0; ++ld_tap; }
        }
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
        // kGeglu: rows 0 .. 159 of the chunk image are W rows 160 g + r (values), rows 160 .. 319 W rows inner + 160 g + (r - 160) (gates)
        // (kPersist: the column group's 160 rows are part of wbase, which changes from tile to tile — the lane offsets do not)
        const uint32_t ggr = kPersist ? 0u : (uint32_t)(gg * kHalf);
        const uint32_t wr = kGeglu ? (r < (uint32_t)kHalf ? ggr + r : (uint32_t)inner + ggr + (r - (uint32_t)kHalf)) : r;
        p_voff[i] = wr * w_row_bytes + 16u * (slot ^ ((r >> 1) & 7u));
    }
    const char* wbase = reinterpret_cast<const char*>(w) + (int64_t)c0 * (kKC * 2) + (kPersist && kGeglu ? (int64_t)gg * kHalf * K * 2 : 0);
    // kPersist: the tile after this one (its x rows, W rows, and where its bias comes from), decoded while this one runs
    const char* nx_xbase = xbase;
    const char* nx_wbase = wbase;
    uint32_t nx_x_voff[2] = {x_voff[0], x_voff[1]};
    int64_t nx_row0 = row0;
    int nx_g = 0;
    bool has_next = false;
    auto decode_next = [&]() __attribute__((always_inline)) {
        const int nt = tile + tile_per;
        has_next = nt < tile_hi;
        if (!has_next) return;
        const int g = nt % cg.groups, rb = nt / cg.groups;               // (column group, row block): as the plain grid's decode
        nx_g = g;
        nx_row0 = (int64_t)rb * kRows + wave * 32;
        const int64_t rb0 = nx_row0 < rows ? nx_row0 : rows - 1;
        nx_xbase = reinterpret_cast<const char*>(x + rb0 * x_rs);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t row = nx_row0 + 16 * t + n16;
            nx_x_voff[t] = (uint32_t)((row < rows ? row - rb0 : 0) * x_rs * 2 + 16 * kg);
        }
        nx_wbase = reinterpret_cast<const char*>(w_all) + (int64_t)g * (kGeglu ? kHalf : kN) * K * 2;
    };
    // wave 0 asks for the next tile's bias (see kBiasLds); lanes past the piece's end read the piece's first bytes again (never used)
    auto stage_next_bias = [&]() __attribute__((always_inline)) {
        if (!bias_all || wave != 0) return;
        const float* const pa = kGeglu ? bias_all + nx_g * kHalf : bias_all + nx_g * kN;
        const float* const pb = kGeglu ? bias_all + inner + nx_g * kHalf : bias_all + nx_g * kN + 256;
        const uint32_t na = kGeglu ? 40u : 64u, nb = kGeglu ? 40u : 16u;             // lanes with 16 valid bytes
        dma_piece(pa, (uint32_t)lane < na ? 16u * (uint32_t)lane : 0u, __builtin_amdgcn_readfirstlane(lds0 + kBiasLds));
        dma_piece(pb, (uint32_t)lane < nb ? 16u * (uint32_t)lane : 0u, __builtin_amdgcn_readfirstlane(lds0 + kBiasLds + 1024u));
    };
    auto acc_from_staged_bias = [&]() __attribute__((always_inline)) {
        MVI_AS3 const float* const sb = (MVI_AS3 const float*)(lds + kBiasLds);
#pragma unroll
        for (int j = 0; j < kNT; ++j) {
            const float b = !bias_all ? 0.f : (kGeglu ? sb[(j < kNT / 2 ? 16 * j : 256 + 16 * (j - kNT / 2)) + n16] : sb[16 * j + n16]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][j][r] = b;
        }
    };
    auto issue_chunk = [&](int c) __attribute__((always_inline)) {
        // chunks past the end re-load the last one (never read): every call issues the same number of pieces, the counted wait stays valid
        const int cc = c < n_chunks ? c : n_chunks - 1;
        const uint32_t slot_off = (uint32_t)((c % kRing) * kChunkBytes);
        const char* const base = wbase + (int64_t)cc * (kKC * 2);
#pragma unroll
        for (int i = 0; i < kPiecesPerLoader; ++i)
            dma_piece(base, p_voff[i], __builtin_amdgcn_readfirstlane(lds0 + slot_off + 1024u * (uint32_t)(wave + i * kLoaders)));
    };

    // ---- LDS read addressing: B operand = W rows; lane (n16, kg), column tile j, k-step s reads row 16 j + n16, 16-byte slot 4 s + kg
    // (at slot ^ ((row >> 1) & 7) = slot ^ (n16 >> 1): the 16 lanes the LDS serves together — four n16 of one kg, eight of the next —
    // land on 16 different 4-bank groups)
    uint32_t ka[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) ka[s] = (uint32_t)(n16 * 128 + (((4 * s + kg) ^ (n16 >> 1)) << 4));
    auto wfrag = [&](uint32_t slot_base, int q) __attribute__((always_inline)) {      // q = 20 s + j
        const int s = q / kNT, j = q % kNT;
        return *reinterpret_cast<MVI_AS3 const u32x4*>(lds + slot_base + ka[s] + j * 2048);
    };

    // One chunk: 40 W fragments, each for the two row tiles = 80 MFMAs (v_mfma_f32_16x16x32), fragments requested kAhead ahead.
    // Forty independent accumulator chains: no MFMA waits for the one before it.
    constexpr int kAhead = 4;
    auto chunk_fn = [&](uint32_t slot_base, u32x4 (&xr)[4], uint32_t (&keep)[2]) __attribute__((always_inline)) {
        if (kConv && __builtin_amdgcn_ballot_w64((keep[0] & keep[1]) == 0u) != 0ull) {
            // the fragment is an output of assembly the compiler believes complete: this statement (volatile, so it stays behind the
            // wait that closed the previous chunk) is what the masking depends on
            asm volatile("" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[i] &= keep[i >> 1];
            asm volatile("s_nop 3" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));      // (vector write -> matrix read, by hand: see Mma)
        }
        u32x4 wf[kAhead + 1];
#pragma unroll
        for (int q = 0; q < kAhead; ++q) wf[q] = wfrag(slot_base, q);
#pragma unroll
        for (int q = 0; q < 2 * kNT; ++q) {
            if (q + kAhead < 2 * kNT) wf[(q + kAhead) % (kAhead + 1)] = wfrag(slot_base, q + kAhead);
            M::mfma(acc[0][q % kNT], xr[q / kNT], wf[q % (kAhead + 1)]);
            M::mfma(acc[1][q % kNT], xr[2 + q / kNT], wf[q % (kAhead + 1)]);
        }
        // (the order above is the order issued: LDS reads do not cross the volatile statements, and the compiler counts lgkmcnt for their
        // operands)
    };
    // Loaders issue the chunk two ahead (its slot held chunk c - 1, which nobody reads any more), then every wave waits for
    // everything older than those pieces — the next chunk's W pieces and its own next x rows among them — and the block meets.
    // Round 5, tried and not kept: the eight waves as two groups half a chunk apart (the barrier at the END of a chunk for waves 0 - 3,
    // in the MIDDLE for waves 4 - 7), so that a SIMD's matrix pipe always has one wave in mid-chunk while the other sits at the barrier
    // and waits for its first fragments. Correct, and 2 - 3 % SLOWER on every shape (profiles/round5_n320_phased_ab.txt): the loop
    // spends 3620 cycles per chunk on 2560 cycles of MFMAs at 1.89 GHz (in-kernel stamps: tools/n320_dev/build_stamped.sh) — the chip trades the idle
    // cycles for clock, and filling them returns as a lower clock, not as time.
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
    uint32_t ka_keep[2], kb_keep[2] = {~0u, ~0u};
    load_x(0, xa, ka_keep);
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
    uint32_t islot = (uint32_t)(2 * kChunkBytes);                    // kPersist: the ring slot of the next chunk to be requested (chunks 0, 1 are out)
    // kPersist forms of load_x / close_chunk: chunk index ci of THIS tile, or — past its end — chunk ci - n_chunks of the next tile (the
    // last tile of the block re-loads its last chunk, as the plain form does). The addresses are selected, never the loads: an
    // asynchronous load inside a branch ends in a register copy of rows that have not landed (profiles/HISTORY.md, round 5).
    auto load_x_p = [&](int ci, u32x4 (&xr)[4]) __attribute__((always_inline)) {
        const bool nxt = ci >= n_chunks && has_next;
        const char* const base = (nxt ? nx_xbase + (int64_t)(ci - n_chunks) * (kKC * 2) : xbase + (int64_t)(ci < n_chunks ? ci : n_chunks - 1) * (kKC * 2));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t voff = nxt ? nx_x_voff[t] : x_voff[t];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) xr[2 * t + s2] = load16_async(base + 64 * s2, voff);
        }
    };
    auto close_chunk_p = [&](int c) __attribute__((always_inline)) {
        if (loader) {
            const int ci = c + 2;
            const bool nxt = ci >= n_chunks && has_next;
            const char* const base = (nxt ? nx_wbase + (int64_t)(ci - n_chunks) * (kKC * 2) : wbase + (int64_t)(ci < n_chunks ? ci : n_chunks - 1) * (kKC * 2));
#pragma unroll
            for (int i = 0; i < kPiecesPerLoader; ++i)
                dma_piece(base, p_voff[i], __builtin_amdgcn_readfirstlane(lds0 + islot + 1024u * (uint32_t)(wave + i * kLoaders)));
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kPiecesPerLoader) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        islot = next_slot(islot);
        __builtin_amdgcn_sched_barrier(0);
    };
    // kPersist, behind a tile's epilogue: the decoded next tile becomes the tile in hand. Its W chunks 0, 1 and x rows of chunk 0 have
    // landed (every wave's wait in front of the epilogue); the barrier makes the loaders' pieces every wave's.
    auto advance_tile = [&]() __attribute__((always_inline)) {
        tile += tile_per;
        xbase = nx_xbase;
        wbase = nx_wbase;
        x_voff[0] = nx_x_voff[0];
        x_voff[1] = nx_x_voff[1];
        row0 = nx_row0;
        out = out_all + nx_g * (kGeglu ? kHalf : kN);
        acc_from_staged_bias();
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
  for (;;) {                                                         // (one pass unless kPersist)
    if (kPersist) {
        decode_next();
        for (int c = 0; c + 1 < n_chunks; c += 2) {                  // (the host takes this form for an even number of chunks only)
            load_x_p(c + 1, xb);
            chunk_fn(slot, xa, ka_keep);
            if (c == 0 && has_next) stage_next_bias();               // (behind the barrier that followed every wave's read of the staged bias)
            close_chunk_p(c);
            slot = next_slot(slot);
            load_x_p(c + 2, xa);
            chunk_fn(slot, xb, kb_keep);
            close_chunk_p(c + 1);
            slot = next_slot(slot);
        }
    } else {
    int c = 0;
    for (; c + 1 < n_chunks; c += 2) {
        load_x(c + 1, xb, kb_keep);                                  // (c + 1 < n_chunks)
        chunk_fn(slot, xa, ka_keep);
        close_chunk(c);
        slot = next_slot(slot);
        load_x(c + 2 < n_chunks ? c + 2 : n_chunks - 1, xa, ka_keep);    // past the end: re-read the last chunk's rows (never used)
        chunk_fn(slot, xb, kb_keep);
        close_chunk(c + 1);
        slot = next_slot(slot);
    }
    if (c < n_chunks) chunk_fn(slot, xa, ka_keep);                   // K / 64 odd: one chunk left
    }
    // trailing (unused) pieces and rows land before the block ends; mfma_settle: the last matrix instructions (8 passes each) have
    // written their accumulators before anything the compiler schedules reads one
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < kNT; ++j) asm volatile("" : "+v"(acc[t][j]));

    if (kSplit) {
        // fp32 partial sums straight from the accumulators: register r of tile (t, j) = row 16 t + 4 kg + r, column 16 j + n16
        const int64_t c_tot = cg.out_cols ? (int64_t)cg.out_cols : (int64_t)cg.groups * kN;
        const int64_t rows_pad = (int64_t)(n_blocks / (cg.groups * cg.ksplit)) * kRows;
        float* const pbase = part + ((int64_t)ks * rows_pad + row0 + 4 * kg) * c_tot + part_col0 + n16;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < kNT; ++j) {
                if (cg.out_cols && part_col0 + 16 * j >= cg.out_cols) continue;        // (the zero rows C_out was padded with)
#pragma unroll
                for (int r = 0; r < 4; ++r) pbase[(16 * t + r) * c_tot + 16 * j] = acc[t][j][r];
            }
        return;
    }
    // ---- outputs leave through the wave's LDS tile [32 rows][64 columns = 128 bytes] — four column tiles per flush — and from there as
    // four 16-byte stores per lane (8 rows x 128 bytes per instruction). The lane writes 2-byte values of rows 16 t + 4 kg + r: the four
    // kg of an instruction would meet in the same banks (rows 4 apart, 128-byte rows), so row R keeps its 16-byte slots at
    // slot ^ (2 (R >> 2 & 3)) and the reader undoes it.
    const uint32_t otile = (uint32_t)(kRing * kChunkBytes + wave * 4096);
    const uint32_t ot_w = otile + (uint32_t)(4 * kg * 128 + (n16 & 7) * 2);       // + 2048 t + 128 r + 16 ((2 jj + (n16 >> 3)) ^ 2 kg)
    const uint32_t ot_w_slot = (uint32_t)(n16 >> 3);
    // reader: lane -> row 8 i + (lane >> 3), logical slot lane & 7; (row >> 2) & 3 = (2 i + (lane >> 5)) & 3
    const uint32_t ot_r_row = otile + (uint32_t)((lane >> 3) * 128);
    auto ot_r = [&](int i) __attribute__((always_inline)) {
        return ot_r_row + 1024u * (uint32_t)i + 16u * ((uint32_t)(lane & 7) ^ (2u * ((uint32_t)(2 * i + (lane >> 5)) & 3u)));
    };
    auto tile_put = [&](int t, int jj, int r, uint32_t pk) __attribute__((always_inline)) {      // rows r and r + 1 of tile (t, jj of the flush)
        const uint32_t a = ot_w + 2048u * (uint32_t)t + 128u * (uint32_t)r + 16u * (((uint32_t)(2 * jj) + ot_w_slot) ^ (2u * (uint32_t)kg));
        *reinterpret_cast<MVI_AS3 uint16_t*>(lds + a) = (uint16_t)(pk & 0xFFFFu);
        *reinterpret_cast<MVI_AS3 uint16_t*>(lds + a + 128) = (uint16_t)(pk >> 16);
    };
    if (kGeglu) {
        // value tile j (columns 16 j ..) and gate tile j + 10 sit in the same lane and register: gate in place, then the ten tiles leave
        // like the plain form's — four tiles (one 128-byte run per row) per flush, the last two alone (64 bytes per row)
        char* const gbase = reinterpret_cast<char*>(out + row0 * o_rs);
        const int64_t grow_bytes = o_rs * 2;
        const int64_t gst_off = (int64_t)(lane >> 3) * grow_bytes + (lane & 7) * 16;
#pragma unroll
        for (int j0 = 0; j0 < kNT / 2; j0 += 4) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (j0 + jj < kNT / 2) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; r += 2)
                            if (cg.dup3) {           // (kGeglu launches carry the packed-erf switch in this otherwise unused field)
                                const f32p o2 = geglu2(f32p{acc[t][j0 + jj][r], acc[t][j0 + jj][r + 1]},
                                                       f32p{acc[t][j0 + jj + kNT / 2][r], acc[t][j0 + jj + kNT / 2][r + 1]});
                                tile_put(t, jj, r, M::pack2(o2.x, o2.y));
                            } else
                            tile_put(t, jj, r, M::pack2(geglu1(acc[t][j0 + jj][r], acc[t][j0 + jj + kNT / 2][r]),
                                                        geglu1(acc[t][j0 + jj][r + 1], acc[t][j0 + jj + kNT / 2][r + 1])));
                }
            }
            char* const op = gbase + (j0 / 4) * 128;
            const bool half = j0 + 2 >= kNT / 2;                  // the last flush holds two tiles: lanes of the upper 64 bytes have nothing
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ot_r(i));
                if (!half || (lane & 7) < 4) *reinterpret_cast<u32x4*>(op + (8 * i) * grow_bytes + gst_off) = v;
            }
        }
        if (!kPersist || !has_next) return;
        advance_tile();
        continue;
    }
    char* const obase = reinterpret_cast<char*>(out + row0 * o_rs);
    const int64_t orow_bytes = o_rs * 2;
    const int64_t st_off = (int64_t)(lane >> 3) * orow_bytes + (lane & 7) * 16;
    if constexpr (kLn) {
        // In the STORE layout: after a flush's four column tiles went through the LDS tile, lane (lr = lane / 8, lc = lane % 8) reads
        // columns 64 f + 8 lc .. + 7 of rows 8 i + lr (i < 4) as 16 bytes — the layout a coalesced load of the residual has too.
        // The row's 320 values stay in registers packed (5 flushes x 4 rows x 4 dwords, while the accumulators they came from die);
        // a row is spread over the 8 lanes of equal lr: mean and centred sum of squares by xor shuffles, two passes over registers as
        // in add_layernorm_kernel.
        const int lr = lane >> 3, lc = lane & 7;
        const T* const resid = (const T*)ln.resid;
        const T* const rowv = (const T*)ln.row;
        int64_t rd[4], rw[4];                                    // element offsets of the lane's rows in resid (clamped) and in row[]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = row0 + 8 * i + lr, rc = r < rows ? r : rows - 1;
            rd[i] = rc * o_rs + 8 * lc;
            rw[i] = rowv ? (rc / ln.row_div) * kN + 8 * lc : 0;
        }
        u32x4 sv[kNT / 4][4];
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < kNT / 4; ++f) {
            // (asking for flush f + 1's pieces a flush ahead, and for all LayerNorm weights before the statistics, was measured 5 %
            // SLOWER: the epilogue is bound by the bytes — all 256 blocks of a round reach it together — not by latency)
            u32x4 rx[4], rr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (resid) rx[i] = *reinterpret_cast<const u32x4*>(resid + rd[i] + 64 * f);
                if (rowv) rr[i] = *reinterpret_cast<const u32x4*>(rowv + rw[i] + 64 * f);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) tile_put(t, jj, r, M::pack2(acc[t][4 * f + jj][r], acc[t][4 * f + jj][r + 1]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 v = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ot_r(i));         // h, rounded
                const int64_t o_off = (8 * i) * orow_bytes + st_off + f * 128;
                if (resid) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = M::pack2(M::lo(rx[i][k]) + M::lo(v[k]), M::hi(rx[i][k]) + M::hi(v[k]));
                }
                if (ln.s_pre) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>((T*)ln.s_pre + row0 * o_rs) + o_off) = v;
                if (rowv) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = M::pack2(M::lo(v[k]) + M::lo(rr[i][k]), M::hi(v[k]) + M::hi(rr[i][k]));
                }
                if (ln.s) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>((T*)ln.s + row0 * o_rs) + o_off) = v;
                sv[f][i] = v;
#pragma unroll
                for (int k = 0; k < 4; ++k) sum[i] += M::lo(v[k]) + M::hi(v[k]);
            }
        }
        float mean[4], rstd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = sum[i];
            t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4);
            mean[i] = t / (float)kN;
            float m2 = 0.f;
#pragma unroll
            for (int f = 0; f < kNT / 4; ++f)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d0 = M::lo(sv[f][i][k]) - mean[i], d1 = M::hi(sv[f][i][k]) - mean[i];
                    m2 = __builtin_fmaf(d0, d0, __builtin_fmaf(d1, d1, m2));
                }
            m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2); m2 += __shfl_xor(m2, 4);
            rstd[i] = rsqrtf(m2 / (float)kN + ln.eps);
        }
#pragma unroll
        for (int f = 0; f < kNT / 4; ++f) {
            const float4* const wp = reinterpret_cast<const float4*>(ln.w + 64 * f + 8 * lc);
            const float4* const bp = reinterpret_cast<const float4*>(ln.b + 64 * f + 8 * lc);
            const float4 w0 = wp[0], w1 = wp[1], b0 = bp[0], b1 = bp[1];
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = M::pack2((M::lo(sv[f][i][k]) - mean[i]) * rstd[i] * wv[2 * k] + bv[2 * k],
                                    (M::hi(sv[f][i][k]) - mean[i]) * rstd[i] * wv[2 * k + 1] + bv[2 * k + 1]);
                *reinterpret_cast<u32x4*>(obase + (8 * i) * orow_bytes + st_off + f * 128) = o;
            }
        }
        return;
    }
    float cs1[kStats ? kNT : 1], cs2[kStats ? kNT : 1];          // kStats: per column tile, sum and sum of squares of this lane's 8 rows
#pragma unroll
    for (int j = 0; j < (kStats ? kNT : 1); ++j) cs1[j] = cs2[j] = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < kNT; j0 += 4) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const uint32_t pk = M::pack2(acc[t][j0 + jj][r], acc[t][j0 + jj][r + 1]);
                    if (kStats) {                                // of the values as stored (the norm reads the rounded tensor)
                        const float a = M::lo(pk), b = M::hi(pk);
                        cs1[j0 + jj] += a + b;
                        cs2[j0 + jj] = __builtin_fmaf(a, a, __builtin_fmaf(b, b, cs2[j0 + jj]));
                    }
                    tile_put(t, jj, r, pk);
                }
            }
        }
        char* const op = obase + (j0 / 4) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v = *reinterpret_cast<MVI_AS3 const u32x4*>(lds + ot_r(i));
            *reinterpret_cast<u32x4*>(op + (8 * i) * orow_bytes + st_off) = v;
        }
    }
    if (kStats && gn.part) {
        // wave partials -> the wave's own 4 KiB output tile (its flush reads are issued: LDS operations of one wave execute in order):
        // [320 channels][2] floats = 2560 bytes; the four kg hold different rows of a column
#pragma unroll
        for (int j = 0; j < kNT; ++j) {
            cs1[j] += __shfl_xor(cs1[j], 16);
            cs2[j] += __shfl_xor(cs2[j], 16);
            cs1[j] += __shfl_xor(cs1[j], 32);
            cs2[j] += __shfl_xor(cs2[j], 32);
        }
        MVI_AS3 float* const wp = (MVI_AS3 float*)(lds + otile);
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < kNT; ++j) { wp[2 * (16 * j + n16)] = cs1[j]; wp[2 * (16 * j + n16) + 1] = cs2[j]; }
        }
        __syncthreads();
        // channel totals over the block's 256 rows -> the spare 1536 bytes behind the partials of tile (channel / 40)
        MVI_AS3 float* const tiles = (MVI_AS3 float*)(lds + kRing * kChunkBytes);
        if (tid < kN) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int wv = 0; wv < kWaves; ++wv) { a += tiles[wv * 1024 + 2 * tid]; b += tiles[wv * 1024 + 2 * tid + 1]; }
            MVI_AS3 float* const tot = tiles + (tid / 40) * 1024 + 640 + 2 * (tid % 40);
            tot[0] = a; tot[1] = b;
        }
        __syncthreads();
        const int c_tot = cg.groups * kN, Cg = c_tot / gn.G, n_g = kN / Cg;      // groups inside this block's 320 channels
        if (tid < n_g) {
            const int64_t r0 = (int64_t)bid * kRows;                              // the block's first row: all 256 belong to one sample
            const int64_t n = r0 / gn.S;
            const int chunk = (int)((r0 - n * gn.S) / kRows), chunks = gn.S / kRows;
            const float* cb = gn.chan_bias ? gn.chan_bias + n * c_tot + part_col0 : nullptr;
            const float rws = (float)kRows;
            float tot = 0.f;
            for (int c = tid * Cg; c < (tid + 1) * Cg; ++c)
                tot += tiles[(c / 40) * 1024 + 640 + 2 * (c % 40)] + rws * (cb ? cb[c] : 0.f);
            const float mean = tot / (rws * Cg);
            float m2 = 0.f;
            for (int c = tid * Cg; c < (tid + 1) * Cg; ++c) {
                const float s1 = tiles[(c / 40) * 1024 + 640 + 2 * (c % 40)], s2 = tiles[(c / 40) * 1024 + 640 + 2 * (c % 40) + 1];
                const float d = mean - (cb ? cb[c] : 0.f);                       // the group mean seen from this channel's own offset
                m2 += s2 - 2.f * d * s1 + rws * d * d;
            }
            float* p = gn.part + ((n * chunks + chunk) * gn.G + part_col0 / Cg + tid) * 3;
            p[0] = rws * Cg; p[1] = mean; p[2] = m2 > 0.f ? m2 : 0.f;
        }
    }
    if (!kPersist || !has_next) return;
    advance_tile();
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

// The persistent form's grid: one block per CU of the current device, rounded down to a multiple of 8 (the XCDs take blocks in turn).
static int persist_grid() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? (prop.multiProcessorCount / 8) * 8 : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}
// Tiles walked by one block per CU with the next tile's first loads under the last chunks of the one in hand (kPersist): for launches of
// more tiles than CUs and an even number of chunks >= 4. MVI_N320_PERSIST=0: the plain grid everywhere (same-box A/B runs).
static bool persist_ok(int64_t n_tiles, int K) {
    const char* const e = getenv("MVI_N320_PERSIST");             // (read per launch: the parity test compares both forms in one process)
    const int g = e && e[0] == '0' ? 0 : persist_grid();
    const int chunks = K / ln3::kKC;
    return g >= 8 && n_tiles > (int64_t)g && n_tiles <= 0x7FFFFFFFll && chunks >= 4 && chunks % 2 == 0;
}

template <typename T, bool kConv = false, bool kSplit = false, bool kStats = false, bool kGeglu = false, bool kLn = false, int NOUT = 320,
          bool kUps = false, bool kPersist = false>
static int linear_n320_launch(const void* x, const void* w, const float* bias, void* out, int64_t rows, int K, int64_t x_rs, int64_t o_rs,
                              hipStream_t st, ln3::ConvGeom cg = {0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0}, float* part = nullptr,
                              ln3::GnStats gn = {nullptr, nullptr, 0, 0},
                              ln3::LnEpi ln = {nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, 0.f}) {
    using namespace ln3;
    const int64_t n_blocks = (rows + kRows - 1) / kRows * cg.groups * cg.ksplit;
    if (n_blocks > 0x7FFFFFFFll) return MVI_EINVAL;
    static unsigned long long attr_set = 0;                      // per device and instantiation: the opt-in for > 64 KiB of dynamic LDS
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MVI_EHIP;
    auto kern = &linear_n320_kernel<T, kConv, kSplit, kStats, kGeglu, kLn, NOUT, kUps, kPersist>;
    constexpr int kLdsBytes = kRing * NOUT * kKC * 2 + kWaves * 4096 + (kPersist ? 2048 : 0);       // (+ the staged bias of the next tile)
    if (!(attr_set >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MVI_EHIP;
        attr_set |= 1ull << dev;
    }
    // kPersist: one block per CU walks the tiles (the caller checked persist_ok: more tiles than CUs, an even number of chunks)
    const unsigned grid = kPersist ? (unsigned)persist_grid() : (unsigned)n_blocks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWaves), kLdsBytes, st, (const T*)x, (const T*)w, bias, (T*)out, rows, K, x_rs,
                       o_rs, (int)n_blocks, cg, part, gn, ln);
    if (kSplit && !cg.out_cols) {
        const int c_tot = cg.groups * kN;
        const int64_t threads = rows * (c_tot / 8), rows_pad = (rows + kRows - 1) / kRows * kRows;
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, (T*)out,
                           rows, c_tot, o_rs, cg.ksplit, rows_pad);
    }
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi

extern "C" int mvi_linear_n320_supported(int32_t K, int32_t out_features, int32_t dtype) {
    return out_features > 0 && out_features % mvi::ln3::kN == 0 && K >= 2 * mvi::ln3::kKC && K % mvi::ln3::kKC == 0 &&
           (dtype == MVI_DT_BF16 || dtype == MVI_DT_F16);
}

extern "C" int mvi_linear_n320(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity,
                               int32_t K, int32_t out_features, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                               void* stream) {
    if (rows < 0 || !mvi_linear_n320_supported(K, out_features, dtype))
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: needs out_features a multiple of 320, K a multiple of 64 (>= 128), bf16 or f16");
    if (rows == 0) return MVI_OK;
    if (!x || !weight || !out) return mvi::unet_fail(MVI_EINVAL, "linear_n320: NULL pointer");
    if (out_rows_capacity < mvi_ff_geglu_out_rows(rows))
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: out needs room for mvi_ff_geglu_out_rows(rows) rows (whole 256-row blocks are stored)");
    if (x_row_stride < K || out_row_stride < out_features || x_row_stride % 8 || out_row_stride % 8 ||
        ((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out) % 16)
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: x, weight and out rows must be 16-byte aligned");
    if ((int64_t)mvi::ln3::kN * K * 2 > 0xFFFFFFFFll || 256 * x_row_stride * 2 > 0xFFFFFFFFll)
        return mvi::unet_fail(MVI_EINVAL, "linear_n320: weight / row block exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    // out_features = 320 g (round 5): g column groups per row block, neighbours in the grid — the blocks of one row block read the same
    // x rows, the first from HBM and the others from L2 (the convolutions' column groups)
    const mvi::ln3::ConvGeom cg = {0, 0, 0, 0, out_features / mvi::ln3::kN, 1, 1, 0, 0, 0, 0};
    // (the persistent form — kPersist, taken by the GEGLU projection below — is 0 ... +5 % SLOWER here: this epilogue is 20 stores, nothing
    // for the next tile's first loads to hide under, and the plain grid's dispatch already overlaps a block's drain with the next one's start:
    // profiles/round6_n320_persistent.txt)
    const int rc = dtype == MVI_DT_BF16 ? mvi::linear_n320_launch<__hip_bfloat16>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st, cg)
                                        : mvi::linear_n320_launch<__half>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st, cg);
    return rc ? mvi::unet_fail(rc, "linear_n320: kernel launch failed") : MVI_OK;
}

// Projection into 320 channels + residual add(s) + LayerNorm in one kernel: see kLn at the kernel
extern "C" int mvi_linear_n320_add_layernorm(const void* x, const void* weight, const float* bias, int64_t rows, int64_t out_rows_capacity,
                                             int32_t K, int64_t x_row_stride, const void* resid, const void* row, int64_t row_div,
                                             const float* ln_weight, const float* ln_bias, float eps, void* s_pre, void* s, void* y,
                                             int64_t out_row_stride, int32_t dtype, void* stream) {
    if (rows < 0 || !mvi_linear_n320_supported(K, mvi::ln3::kN, dtype))
        return mvi::unet_fail(MVI_EINVAL, "linear_n320_add_layernorm: needs K a multiple of 64 (>= 128), bf16 or f16 (320 outputs)");
    if (rows == 0) return MVI_OK;
    if (!x || !weight || !y || !ln_weight || !ln_bias) return mvi::unet_fail(MVI_EINVAL, "linear_n320_add_layernorm: NULL pointer");
    if (row && row_div <= 0) return mvi::unet_fail(MVI_EINVAL, "linear_n320_add_layernorm: row_div must be positive");
    if (out_rows_capacity < mvi_ff_geglu_out_rows(rows))
        return mvi::unet_fail(MVI_EINVAL, "linear_n320_add_layernorm: the outputs need room for mvi_ff_geglu_out_rows(rows) rows");
    if (x_row_stride < K || out_row_stride < mvi::ln3::kN || x_row_stride % 8 || out_row_stride % 8 ||
        ((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)resid | (uintptr_t)row | (uintptr_t)s_pre | (uintptr_t)s |
         (uintptr_t)ln_weight | (uintptr_t)ln_bias) % 16)
        return mvi::unet_fail(MVI_EINVAL, "linear_n320_add_layernorm: every row (and the LayerNorm vectors) must be 16-byte aligned");
    if ((int64_t)mvi::ln3::kN * K * 2 > 0xFFFFFFFFll || 256 * x_row_stride * 2 > 0xFFFFFFFFll)
        return mvi::unet_fail(MVI_EINVAL, "linear_n320_add_layernorm: weight / row block exceeds 32-bit byte offsets");
    const mvi::ln3::LnEpi ln = {resid, row, row ? row_div : 1, ln_weight, ln_bias, s_pre, s, eps};
    const mvi::ln3::ConvGeom cg = {0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == MVI_DT_BF16
                       ? mvi::linear_n320_launch<__hip_bfloat16, false, false, false, false, true>(x, weight, bias, y, rows, K, x_row_stride,
                                                                                                   out_row_stride, st, cg, nullptr, {nullptr, nullptr, 0, 0}, ln)
                       : mvi::linear_n320_launch<__half, false, false, false, false, true>(x, weight, bias, y, rows, K, x_row_stride, out_row_stride, st,
                                                                                           cg, nullptr, {nullptr, nullptr, 0, 0}, ln);
    return rc ? mvi::unet_fail(rc, "linear_n320_add_layernorm: kernel launch failed") : MVI_OK;
}

// GEGLU projection with a long contraction (the level-1 / level-2 FeedForward layers): see kGeglu at the kernel
extern "C" int mvi_ff_geglu_n320_supported(int32_t K, int32_t inner, int32_t dtype) {
    return K >= 2 * mvi::ln3::kKC && K % mvi::ln3::kKC == 0 && inner > 0 && inner % (mvi::ln3::kN / 2) == 0 &&
           (int64_t)2 * inner * K * 2 <= 0xFFFFFFFFll && (dtype == MVI_DT_BF16 || dtype == MVI_DT_F16);
}

extern "C" int mvi_ff_geglu_n320(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity,
                                 int32_t K, int32_t inner, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, void* stream) {
    if (rows < 0 || !mvi_ff_geglu_n320_supported(K, inner, dtype))
        return mvi::unet_fail(MVI_EINVAL, "ff_geglu_n320: needs K a multiple of 64 (>= 128), inner a multiple of 160, bf16 or f16");
    if (rows == 0) return MVI_OK;
    if (!x || !weight || !out) return mvi::unet_fail(MVI_EINVAL, "ff_geglu_n320: NULL pointer");
    if (out_rows_capacity < mvi_ff_geglu_out_rows(rows))
        return mvi::unet_fail(MVI_EINVAL, "ff_geglu_n320: out needs room for mvi_ff_geglu_out_rows(rows) rows (whole 256-row blocks are stored)");
    if (x_row_stride < K || out_row_stride < inner || x_row_stride % 8 || out_row_stride % 8 ||
        ((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out) % 16)
        return mvi::unet_fail(MVI_EINVAL, "ff_geglu_n320: x, weight and out rows must be 16-byte aligned");
    if (256 * x_row_stride * 2 > 0xFFFFFFFFll) return mvi::unet_fail(MVI_EINVAL, "ff_geglu_n320: row block exceeds 32-bit byte offsets");
    const mvi::ln3::ConvGeom cg = {0, 0, 0, 0, inner / (mvi::ln3::kN / 2), 1, 1, 0, 0, 0, 0, mvi::ln3::g_geglu_packed, 0};
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_tiles = (rows + mvi::ln3::kRows - 1) / mvi::ln3::kRows * cg.groups;
    int rc;
    if (mvi::persist_ok(n_tiles, K))
        rc = dtype == MVI_DT_BF16
                 ? mvi::linear_n320_launch<__hip_bfloat16, false, false, false, true, false, 320, false, true>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st, cg)
                 : mvi::linear_n320_launch<__half, false, false, false, true, false, 320, false, true>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st, cg);
    else
        rc = dtype == MVI_DT_BF16
                 ? mvi::linear_n320_launch<__hip_bfloat16, false, false, false, true>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st, cg)
                 : mvi::linear_n320_launch<__half, false, false, false, true>(x, weight, bias, out, rows, K, x_row_stride, out_row_stride, st, cg);
    return rc ? mvi::unet_fail(rc, "ff_geglu_n320: kernel launch failed") : MVI_OK;
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

// Order of the 3x3 convolution's contraction (round 5). 0: (tap, channel) — weight [C_out][9][C_in], the form of rounds 3 - 4.
// 1 (default): (64-channel chunk, tap, channel) — weight [C_out][C_in / 64][9][64]: the nine taps of one channel chunk follow each
// other, and they read the SAME 128-byte slices of the block's four image rows shifted by a pixel or a row — 64 KB per block and
// chunk, which stays in the XCD's L2 — where the tap-major order came back to a pixel after 10 - 40 chunks of other data and found it
// evicted: the counters showed the 330 MB activation of a level-0 convolution fetched nine times (3.06 GB per launch at
// 28 x 72x128, 640 -> 320: profiles/round5_pmc_svd_traffic_tap_major.txt). The caller packs the weight in the same order
// (svd/hip_ops.py conv3x3_n320_weight reads mvi_conv3x3_n320_k_order()); the (3,1,1) form keeps (tap, channel).
static int g_conv_k_order = [] { const char* e = getenv("MVI_CONV_K_ORDER"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int mvi_conv3x3_n320_k_order(int32_t set) {
    if (set == 0 || set == 1) g_conv_k_order = set;
    return g_conv_k_order;
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

// Can a convolution launch of this shape leave the statistics of the GroupNorm behind it (kStats)? Unsplit launches whose 256-row
// blocks never straddle two samples, GroupNorm groups that tile the 320 channels of a block.
static int conv_gnstats_ok(int64_t rows, int32_t taps, int32_t stride, int32_t C_in, int32_t C_out, int64_t spatial, int32_t groups) {
    if (stride != 1 || spatial <= 0 || spatial % mvi::ln3::kRows || rows % spatial || groups <= 0 || C_out % groups) return 0;
    const int Cg = C_out / groups;
    if (Cg > 40 || mvi::ln3::kN % Cg) return 0;                   // (the block's per-channel totals sit 40 to an LDS tile)
    return conv_ksplit(rows, taps, C_in, C_out) == 1;
}

static int conv_taps_n320(const char* what, const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t H, int32_t W,
                          int32_t taps, int32_t stride, int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride,
                          int32_t dtype, void* workspace, size_t workspace_bytes, void* stream,
                          mvi::ln3::GnStats gn = {nullptr, nullptr, 0, 0}, int32_t ups = 0) {
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
    mvi::ln3::ConvGeom cg = {H, W, C_in / mvi::ln3::kKC, taps, C_out / mvi::ln3::kN, ks, stride, Ho, Wo, taps == 9 ? g_conv_k_order : 0,
                             (taps == 3 && g_conv_k_order && W % mvi::ln3::kRows == 0) ? 1 : 0};
    int rc;
    if (ups) {                                       // (H, W: the upsampled image — both even by construction; its own instantiation, never K-split)
        if (taps != 9 || stride != 1 || gn.part || (H & 1) || (W & 1)) return fail("upsampling: 3x3, stride 1, no statistics");
        cg.ksplit = 1;
        rc = dtype == MVI_DT_BF16
                 ? mvi::linear_n320_launch<__hip_bfloat16, true, false, false, false, false, 320, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st, cg)
                 : mvi::linear_n320_launch<__half, true, false, false, false, false, 320, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st, cg);
    } else if (gn.part) {
        if (!conv_gnstats_ok(rows, taps, stride, C_in, C_out, gn.S, gn.G)) return fail("this shape cannot leave GroupNorm statistics (mvi_conv_n320_gnstats_supported)");
        rc = dtype == MVI_DT_BF16 ? mvi::linear_n320_launch<__hip_bfloat16, true, false, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st, cg, nullptr, gn)
                                  : mvi::linear_n320_launch<__half, true, false, true>(x, weight, bias, out, rows, taps * C_in, C_in, out_row_stride, st, cg, nullptr, gn);
    } else if (ks > 1)
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

// conv3x3(F.interpolate(x, scale_factor=2, mode="nearest")) for token-major x [N, h w, C_in] -> out [N, (2 h)(2 w), C_out]: Upsample.conv
// (openaimodel.py:107-150) with the upsampling in the kernel's addressing — the 4x tensor is neither written nor read. Workspace as for
// mvi_conv3x3_n320 at (N, 2 h, 2 w).
extern "C" int mvi_conv3x3_up2_n320(const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t h, int32_t w,
                                    int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (h <= 0 || w <= 0 || h > (1 << 29) || w > (1 << 29)) return mvi::unet_fail(MVI_EINVAL, "conv3x3_up2_n320: bad image size");
    return conv_taps_n320("conv3x3_up2_n320", x, weight, bias, out, N, 2 * h, 2 * w, 9, 1, C_in, C_out, out_rows_capacity, out_row_stride, dtype,
                          workspace, workspace_bytes, stream, {nullptr, nullptr, 0, 0}, 1);
}

extern "C" int mvi_conv_n320_gnstats_supported(int64_t rows, int32_t taps, int32_t stride, int32_t C_in, int32_t C_out, int64_t spatial,
                                               int32_t groups) {
    return (taps == 9 || taps == 3) && C_out > 0 && C_out % mvi::ln3::kN == 0 && C_in >= mvi::ln3::kKC && C_in % mvi::ln3::kKC == 0
           ? conv_gnstats_ok(rows, taps, stride, C_in, C_out, spatial, groups) : 0;
}

extern "C" size_t mvi_conv_n320_gnstats_bytes(int64_t samples, int64_t spatial, int32_t groups) {
    return samples <= 0 || spatial <= 0 || groups <= 0 ? 0 : (size_t)samples * (size_t)(spatial / mvi::ln3::kRows) * (size_t)groups * 3 * sizeof(float);
}

extern "C" int mvi_conv3x3_n320_gnstats(const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t H, int32_t W,
                                        int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                                        const float* gn_chan_bias, int32_t gn_groups, float* gn_part, size_t gn_part_bytes, void* stream) {
    if (!gn_part || gn_part_bytes < mvi_conv_n320_gnstats_bytes(N, (int64_t)H * W, gn_groups))
        return mvi::unet_fail(MVI_EINVAL, "conv3x3_n320_gnstats: statistics buffer missing or too small (mvi_conv_n320_gnstats_bytes)");
    return conv_taps_n320("conv3x3_n320_gnstats", x, weight, bias, out, N, H, W, 9, 1, C_in, C_out, out_rows_capacity, out_row_stride, dtype,
                          nullptr, 0, stream, mvi::ln3::GnStats{gn_part, gn_chan_bias, gn_groups, H * W});
}

extern "C" int mvi_conv3t_n320_gnstats(const void* x, const void* weight, const float* bias, void* out, int64_t B, int32_t T, int32_t pixels,
                                       int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                                       const float* gn_chan_bias, int32_t gn_groups, float* gn_part, size_t gn_part_bytes, void* stream) {
    if (!gn_part || gn_part_bytes < mvi_conv_n320_gnstats_bytes(B * T, pixels, gn_groups))
        return mvi::unet_fail(MVI_EINVAL, "conv3t_n320_gnstats: statistics buffer missing or too small (mvi_conv_n320_gnstats_bytes)");
    return conv_taps_n320("conv3t_n320_gnstats", x, weight, bias, out, B, T, pixels, 3, 1, C_in, C_out, out_rows_capacity, out_row_stride, dtype,
                          nullptr, 0, stream, mvi::ln3::GnStats{gn_part, gn_chan_bias, gn_groups, pixels});
}

extern "C" int mvi_conv3t_n320(const void* x, const void* weight, const float* bias, void* out, int64_t B, int32_t T, int32_t pixels,
                               int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return conv_taps_n320("conv3t_n320", x, weight, bias, out, B, T, pixels, 3, 1, C_in, C_out, out_rows_capacity, out_row_stride, dtype,
                          workspace, workspace_bytes, stream);
}

// ---- Round 6: convolutions at fp32 ACCURACY on the bf16 matrix pipe — the first-stage (VAE) decoder's 128 / 256 / 512-channel 3x3 and
// (3,1,1) convolutions, which the reference runs in fp32 (`disable_first_stage_autocast: True`, configs/test/svd_f_est_ctrl_simp1.yaml:6;
// sgm/models/diffusion.py:194-212; sgm/modules/diffusionmodules/model.py:604-748; sgm/modules/autoencoding/temporal_ae.py:291-347).
// Every fp32 operand is split into two bf16 values, v = hi + lo (hi = round(v), lo = round(v - hi): 16 mantissa bits together), and
//     x . w  ~=  x_hi . w_hi  +  x_hi . w_lo  +  x_lo . w_hi          (fp32 accumulate; the dropped x_lo . w_lo term is 2^-16 of a product)
// runs as ONE implicit GEMM of this kernel with a three times longer contraction: logical channel axis (hi | hi | lo) of the
// activations against (w_hi | w_lo | w_hi) of the weights. The activations are stored once, as x2 [rows, 2 C] = (hi | lo) bf16 (what
// mvi_groupnorm_silu_tok2tok_split writes: the bytes of the fp32 tensor), and the K loop reads the hi half twice (ConvGeom::dup3).
// The fp32 accumulators are the result (ConvGeom::out_cols): out [padded rows, C_out] fp32, no bias (the callers fold it into the
// next norm / the residual add, as the fp32 path of svd/vae.py does).
// terms = 1: the same launch on ONE rounded value per operand (x [rows, C] bf16 or f16, weight [C_out_padded][taps x C]): products of
// rounded operands accumulated in fp32 with an fp32 result — the arithmetic of an autocast convolution, for the opt-in
// reduced-precision decode (svd/vae.py decode_first_stage(dtype=...)), whose residual stream and norms then stay fp32 like the default's.
//   weight: [C_out_padded][taps x 3 C] bf16 in this kernel's K order (svd/hip_ops.py split3_weight), C_out_padded = a whole number of
//   column groups of mvi_conv_split3_group(C_out) columns (320 when C_out is a multiple of 320, 128 up to 128 channels, else 256), the
//   padding rows zero.
// (128: the 128-channel level of the decoder — a 256-column block would compute 128 columns of zeros)
static int g_split3_group128 = [] { const char* e = getenv("MVI_SPLIT3_GROUP128"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int mvi_conv_split3_group(int32_t C_out) {
    if (C_out > 0 && C_out % 320 == 0) return 320;
    return (g_split3_group128 && C_out > 0 && C_out <= 128) ? 128 : 256;
}

extern "C" int64_t mvi_conv_split3_out_rows(int64_t rows) { return (rows + mvi::ln3::kRows - 1) / mvi::ln3::kRows * mvi::ln3::kRows; }

static int conv_split3(const char* what, const void* x2, const void* weight, float* out, int64_t N, int32_t H, int32_t W, int32_t taps,
                       int32_t C, int32_t C_out, int32_t terms, int32_t dtype, int64_t out_rows_capacity, void* stream) {
    char msg[200];
    auto fail = [&](const char* m) {
        snprintf(msg, sizeof msg, "%s: %s", what, m);
        return mvi::unet_fail(MVI_EINVAL, msg);
    };
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || C % 64 || C_out <= 0 || C_out % 16) return fail("needs C a multiple of 64 and C_out a multiple of 16");
    if (!((terms == 3 && dtype == MVI_DT_BF16) || (terms == 1 && (dtype == MVI_DT_BF16 || dtype == MVI_DT_F16))))
        return fail("terms = 3 (split operands, bf16) or terms = 1 (plain bf16 / f16 operands)");
    if (terms == 1 && taps * C < 2 * mvi::ln3::kKC) return fail("the contraction needs at least two chunks of 64");
    const int64_t rows = N * H * W;
    if (rows == 0) return MVI_OK;
    if (!x2 || !weight || !out) return fail("NULL pointer");
    if (((uintptr_t)x2 | (uintptr_t)weight | (uintptr_t)out) % 16) return fail("x2, weight and out must be 16-byte aligned");
    if (out_rows_capacity < mvi_conv_split3_out_rows(rows)) return fail("out needs room for mvi_conv_split3_out_rows(rows) rows (whole 256-row blocks are stored)");
    const int group = mvi_conv_split3_group(C_out);
    const int groups = (C_out + group - 1) / group;
    const int64_t x_rs = (terms == 3 ? 2 : 1) * (int64_t)C;       // physical row: (hi | lo), or the one rounded value
    if ((int64_t)group * taps * terms * C * 2 > 0xFFFFFFFFll || rows * x_rs * 2 > 0xFFFFFFFFll || (int64_t)H * W > 0x7FFFFFFFll)
        return fail("weight group / activation tensor exceeds 32-bit byte offsets (split the batch)");
    const int k_order = taps == 9 ? g_conv_k_order : 0;
    const mvi::ln3::ConvGeom cg = {H, W, terms * C / mvi::ln3::kKC, taps, groups, 1, 1, H, W, k_order,
                                   (taps == 3 && g_conv_k_order && W % mvi::ln3::kRows == 0) ? 1 : 0, terms == 3 ? 1 : 0, C_out};
    hipStream_t st = (hipStream_t)stream;
    const int K = taps * terms * C;
    int rc;
#define MVI_SPLIT3_LAUNCH(T, N) mvi::linear_n320_launch<T, true, true, false, false, false, N>(x2, weight, nullptr, nullptr, rows, K, x_rs, 0, st, cg, out)
    if (dtype == MVI_DT_BF16)
        rc = group == 320 ? MVI_SPLIT3_LAUNCH(__hip_bfloat16, 320) : group == 256 ? MVI_SPLIT3_LAUNCH(__hip_bfloat16, 256) : MVI_SPLIT3_LAUNCH(__hip_bfloat16, 128);
    else
        rc = group == 320 ? MVI_SPLIT3_LAUNCH(__half, 320) : group == 256 ? MVI_SPLIT3_LAUNCH(__half, 256) : MVI_SPLIT3_LAUNCH(__half, 128);
#undef MVI_SPLIT3_LAUNCH
    if (rc) {
        snprintf(msg, sizeof msg, "%s: kernel launch failed", what);
        return mvi::unet_fail(rc, msg);
    }
    return MVI_OK;
}

extern "C" int mvi_conv3x3_split3_f32(const void* x2, const void* weight, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t C_out,
                                      int32_t terms, int32_t dtype, int64_t out_rows_capacity, void* stream) {
    return conv_split3("conv3x3_split3_f32", x2, weight, out, N, H, W, 9, C, C_out, terms, dtype, out_rows_capacity, stream);
}

extern "C" int mvi_conv3t_split3_f32(const void* x2, const void* weight, float* out, int64_t B, int32_t T, int32_t pixels, int32_t C,
                                     int32_t C_out, int32_t terms, int32_t dtype, int64_t out_rows_capacity, void* stream) {
    return conv_split3("conv3t_split3_f32", x2, weight, out, B, T, pixels, 3, C, C_out, terms, dtype, out_rows_capacity, stream);
}
