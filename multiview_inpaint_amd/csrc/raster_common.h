// Shared declarations of the gfx950 rasterizer kernels (device helpers + scratch layouts).
// Plug-in being replaced: diff_gaussian_rasterization, called at
// gs-simp/gaussian_renderer/__init__.py:85-93. Written for CDNA4 only (wave64, 256-thread blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/mvi_raster.h"

namespace mvi {

constexpr int kTile = MVI_TILE;            // 16x16 pixels per tile, one 256-thread block (4 wave64)
constexpr int kBlock = 256;
constexpr int kPB = 64;                    // threads (= Gaussians) per block of the per-Gaussian preprocess kernels: their LDS
                                           // rows (192 B of SH per Gaussian) cap residency at ~9 waves per CU, and one-wave
                                           // blocks overlap their load / compute / store phases best (measured, backward
                                           // kernel: 256 threads 0.203 ms, 128: 0.202, 64: 0.184; forward 0.134 / 0.122 / 0.125)
constexpr float kNearZ = 0.2f;             // view-space z cull
constexpr float kLowpass = 0.3f;           // cov2D diagonal dilation
constexpr float kFrustumClamp = 1.3f;
constexpr float kLambdaFloor = 0.1f;
constexpr float kAlphaMax = 0.99f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTEps = 0.0001f;
constexpr float kDepthSentinel = 15.0f;    // gs-simp/gen_seq.py:50

constexpr int kSortItems = 8;              // pairs per thread of a radix scatter block
constexpr int kSortWavesP = 4;             // waves per scatter block: 2048-pair tiles for the P-sized level-1 passes,
constexpr int kSortWavesD = 8;             //                          4096-pair tiles for the D-sized level-2 passes
constexpr int kCountTiles = 4;             // sort tiles per block of the count kernel: one 16-byte run per histogram row
__host__ __device__ inline int sort_blocks(int64_t n, int waves) {   // radix-pass blocks, padded so the count kernel writes whole runs
    const int tile = 512 * waves;
    int b = (int)((n + tile - 1) / tile);
    return (b + kCountTiles - 1) / kCountTiles * kCountTiles;
}

// Per-call constants every kernel needs, passed by value (lives in SGPRs).
struct Frame {
    int P, M, deg, W, H, gx, gy;
    float tanfovx, tanfovy, fx, fy, scale_modifier;
    const float* view;   // [16] device
    const float* proj;   // [16] device
    const float* campos; // [3]  device
    const float* bg;     // [3]  device
    // raw-parameter mode (mvi_raster_*_raw): `shs` is features_dc [P,1,3], shs_rest is features_rest [P,M-1,3], and
    // opacities / scales / rotations are the un-activated parameters (sigmoid / exp / normalize happen in the kernels)
    const float* shs_rest = nullptr;
    int raw = 0;
    int bin_v2 = 0;      // binning version 2 (rectangle-expanding partition, raster_binning2.hip): set by make_frame
};

// ---- binning version 2 (raster_binning2.hip): tile grids of at most 256 x 256 tiles (images up to 4096 x 4096) ------------
constexpr int kDsThreads = 256;                     // depth sort: 256 threads x 8 keys = 2048-key tiles
constexpr int kDsItems = 8;
constexpr int kDsTile = kDsThreads * kDsItems;
constexpr int kDsSuper = 16;                        // count-table rows per super row (two-level predecessor sums)
constexpr int kExThreads = 512;                     // expanding partition passes: 512 threads x 4 items
constexpr int kExItems = 4;
constexpr int kExChunk = kExThreads * kExItems;     // items (Gaussians in pass 1, column segments in pass 2) per block
bool binning_v2_enabled();                          // false when MVI_BINNING_LEGACY=1 or after set_binning_version(1)
void set_dev_stamps(int pass, void* buf);            // diagnostics: shader-clock stamps of the partition kernel's phases
int set_binning_version(int v);                     // 1 | 2 (anything else: query only); returns the previous version
inline bool binning_v2_ok(int gx, int gy) { return gx <= 256 && gy <= 256 && binning_v2_enabled(); }

// ---- scratch layouts (all offsets 256-B aligned) ---------------------------------------------
struct GeomView {
    float* depths;            // [P]
    float2* xy;               // [P]
    float4* cov_a;            // [P] cov3D xx, xy, xz, yy   (16-byte and 8-byte records: every lane's
    float2* cov_b;            // [P] cov3D yz, zz           store/load is one whole aligned access)
    float4* conic_opacity;    // [P]
    float4* rgbd;             // [P] r, g, b, view-space depth (one gather per staged list entry)
    uint32_t* tiles_touched;  // [P]
    uint2* rect;              // [P] tile rectangle (x0 | y0 << 16, w | h << 16); w * h = tiles touched, 0 when culled:
                              //     pair emission gathers this one 8-byte record per depth-ordered Gaussian
    uint8_t* clamped;         // [P] bit c set = colour channel c was clamped at 0
    uint32_t* touched_list;   // [P] the touched Gaussians, compacted (any order); touched_count[0] of them
    uint32_t* touched_count;  // [1]
    uint8_t* touched;         // [P] 1 = the render backward added something to this Gaussian's accumulation row (gradient
                              //     support: ~3 % of the Gaussians of the bench scene; the others are occluded). Zeroed with the rows.
    uint32_t* block_sums;     // [npre]   tiles touched per preprocess block (kPB Gaussians)
    uint32_t* block_offsets;  // [npre+1] exclusive scan of block_sums; [npre] = D
    // depth sort of the Gaussians (binning level 1): ping-pong (depth bits, index) pairs
    uint32_t* dkeys[2];       // [P]
    uint32_t* dvals[2];       // [P]; after the 4 passes dvals[0] = Gaussian indices in depth order
    uint32_t* dhist;          // [256 * nsortP]
    uint32_t* dtot;           // [256]
    uint32_t* perm_sums;      // [nblk]   tiles touched per 256 depth-ordered Gaussians
    uint32_t* perm_offsets;   // [nblk+1]
    // binning version 2
    uint2* rect_sorted;       // [P] the tile rectangles in depth order (written by the column count, read by pass 1)
    uint32_t* col_table;      // [256][nblk1] column-major: column segments per (column, block of kExChunk Gaussians)
    uint32_t* col_tot;        // [256] column segments per tile column
    uint32_t* seg_sums;       // [npre] column segments (sum of rectangle widths) per preprocess block
    uint32_t* chunk_first;    // [257] first pass-2 chunk of tile column x; [gx] = number of chunks
    uint32_t* col_start;      // [257] first column segment of tile column x; [gx] = number of segments
    uint32_t* arrivals;       // [1] blocks of the column scan that have finished (the last one builds the chunk table)
    uint32_t* ds_table;       // [nds][256] depth-sort digit counts per 4096-key tile (aliases dhist)
    uint32_t* ds_super[2];    // [nsuper][256] sums over kDsSuper tiles, one region per pass parity (inside dhist)
    int nds, nsuper;
    int nblk1;                // blocks of pass 1
    int nsortP;
    size_t bytes;
};
struct ImageView {
    uint32_t* ranges;     // [2*tiles]
    float* final_T;       // [H*W]
    uint32_t* n_contrib;  // [H*W]
    size_t bytes;
};
struct BinningView {
    // Version 2 (v2 != 0, grids up to 256 x 256 tiles): keys[0] = (first row | rows - 1 << 8) of each column segment,
    // vals[0] = its Gaussian (pass 1 output, in column order); keys[1] = tile id of each sorted pair, vals[1] = the sorted
    // point list (pass 2 output); passes = 1 so that keys / vals[passes & 1] is the result in both versions.
    // block_hist = [256][nsort] rows-per-chunk table, digit_tot = pairs per tile row, col_rel = [gy][gx + 1].
    void* keys[2];        // ping-pong [D] tile ids, key_bytes each (the 64-bit (tile|depth) key is implicit: pairs are
                          // emitted in depth order and stably partitioned by tile)
    int key_bytes;        // 2 while the image has at most 65536 tiles, else 4
    uint32_t* vals[2];    // ping-pong [D] Gaussian indices
    uint32_t* block_hist; // [256 * nsort]  digit-major per-block digit counts / offsets
    uint32_t* digit_tot;  // [256]
    int nsort;            // blocks per radix pass
    int passes;           // 8-bit passes over the tile bits; sorted pairs end in keys/vals[passes & 1]
    int key_bits;         // tile bits
    int v2;               // binning version 2 layout
    uint32_t* col_rel;    // v2: [gy][gx + 1] pairs of tile row y in front of column x (row-relative)
    size_t bytes;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline GeomView carve_geom(void* base, int P) {
    GeomView g;
    char* p = (char*)base;
    size_t o = 0;
    size_t n = (size_t)(P > 0 ? P : 1);
    int nblk = (int)((n + kBlock - 1) / kBlock);
    auto take = [&](size_t b) { char* r = p ? p + o : nullptr; o += align256(b); return r; };
    g.depths = (float*)take(4 * n);
    g.xy = (float2*)take(8 * n);
    g.cov_a = (float4*)take(16 * n);
    g.cov_b = (float2*)take(8 * n);
    g.conic_opacity = (float4*)take(16 * n);
    g.rgbd = (float4*)take(16 * n);
    g.tiles_touched = (uint32_t*)take(4 * n);
    g.rect = (uint2*)take(8 * n);
    g.clamped = (uint8_t*)take(n);
    g.touched = (uint8_t*)take(n);
    g.touched_list = (uint32_t*)take(4 * n);
    g.touched_count = (uint32_t*)take(4);
    const size_t npre = (n + kPB - 1) / kPB;
    g.block_sums = (uint32_t*)take(4 * npre);
    g.block_offsets = (uint32_t*)take(4 * (npre + 1));
    g.nsortP = sort_blocks(n, kSortWavesP);
    for (int i = 0; i < 2; ++i) { g.dkeys[i] = (uint32_t*)take(4 * n); g.dvals[i] = (uint32_t*)take(4 * n); }
    g.dhist = (uint32_t*)take(4 * 256 * (size_t)g.nsortP);
    g.dtot = (uint32_t*)take(4 * 256);
    g.perm_sums = (uint32_t*)take(4 * (size_t)nblk);
    g.perm_offsets = (uint32_t*)take(4 * (size_t)(nblk + 1));
    g.rect_sorted = (uint2*)take(8 * n);
    g.nblk1 = (int)((n + kExChunk - 1) / kExChunk);
    g.col_table = (uint32_t*)take(4 * 256 * (size_t)g.nblk1);
    g.col_tot = (uint32_t*)take(4 * 256);
    g.seg_sums = (uint32_t*)take(4 * npre);
    g.chunk_first = (uint32_t*)take(4 * 257);
    g.col_start = (uint32_t*)take(4 * 257);
    g.arrivals = (uint32_t*)take(4);
    g.nds = (int)((n + kDsTile - 1) / kDsTile);
    g.nsuper = (g.nds + kDsSuper - 1) / kDsSuper;
    g.ds_table = (uint32_t*)take(4 * 256 * (size_t)g.nds);
    g.ds_super[0] = (uint32_t*)take(4 * 256 * (size_t)g.nsuper);
    g.ds_super[1] = (uint32_t*)take(4 * 256 * (size_t)g.nsuper);
    g.bytes = o;
    return g;
}
inline ImageView carve_image(void* base, int W, int H) {
    ImageView v;
    char* p = (char*)base;
    size_t o = 0;
    size_t tiles = (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
    size_t px = (size_t)W * H;
    auto take = [&](size_t b) { char* r = p ? p + o : nullptr; o += align256(b); return r; };
    v.ranges = (uint32_t*)take(8 * (tiles ? tiles : 1));
    v.final_T = (float*)take(4 * (px ? px : 1));
    v.n_contrib = (uint32_t*)take(4 * (px ? px : 1));
    v.bytes = o;
    return v;
}
inline int tile_bits(int W, int H) {
    unsigned tiles = (unsigned)(((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile));
    int b = 0;
    while ((1u << b) < tiles && b < 31) ++b;
    return b > 0 ? b : 1;
}
inline BinningView carve_binning(void* base, int64_t D, int W, int H) {
    BinningView v;
    char* p = (char*)base;
    size_t o = 0;
    size_t n = (size_t)(D > 0 ? D : 1);
    v.nsort = sort_blocks(n, kSortWavesD);
    v.key_bits = tile_bits(W, H);
    v.passes = (v.key_bits + 7) / 8;
    auto take = [&](size_t b) { char* r = p ? p + o : nullptr; o += align256(b); return r; };
    const unsigned tiles = (unsigned)(((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile));
    v.key_bytes = tiles <= 65536u ? 2 : 4;
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    v.v2 = binning_v2_ok(gx, gy) ? 1 : 0;
    v.col_rel = nullptr;
    if (v.v2) {
        // chunks of pass 2: every tile column owns at least one, so at most D / kExChunk + gx (column segments <= pairs);
        // the launches use the exact segment count when the forward knows it
        v.nsort = (int)((n / kExChunk + (size_t)gx + 1 + 3) / 4 * 4);
        v.passes = 1;
        v.key_bytes = 2;
    }
    v.keys[0] = take((size_t)v.key_bytes * n);
    v.keys[1] = take((size_t)v.key_bytes * n);
    v.vals[0] = (uint32_t*)take(4 * n);
    v.vals[1] = (uint32_t*)take(4 * n);
    v.block_hist = (uint32_t*)take(4 * 256 * (size_t)v.nsort);
    v.digit_tot = (uint32_t*)take(4 * 256);
    if (v.v2) v.col_rel = (uint32_t*)take(4 * (size_t)gy * (size_t)(gx + 1));
    v.bytes = o;
    return v;
}

// ---- device helpers ---------------------------------------------------------------------------
// Arithmetic contract shared with oracle/raster_oracle.c: explicit fma chains, no other fusion in
// the functions that feed integer outputs (those are compiled under `#pragma clang fp contract(off)`).
__device__ __forceinline__ float affine3(float m0, float m1, float m2, float m3, float x, float y, float z) {
    return __builtin_fmaf(m2, z, __builtin_fmaf(m1, y, __builtin_fmaf(m0, x, m3)));
}
__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
#pragma clang fp contract(off)
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
}

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& x0,
                                          int& y0, int& x1, int& y1) {
#pragma clang fp contract(off)
    x0 = min(gx, max(0, (int)((px - (float)radius) / (float)kTile)));
    y0 = min(gy, max(0, (int)((py - (float)radius) / (float)kTile)));
    x1 = min(gx, max(0, (int)((px + (float)radius + (float)(kTile - 1)) / (float)kTile)));
    y1 = min(gy, max(0, (int)((py + (float)radius + (float)(kTile - 1)) / (float)kTile)));
}

// Sum over the 64 lanes of a wave with DPP moves; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3
    return v;
}
// Sum over each 16-lane DPP row; the row total lands in lanes 15, 31, 47, 63.
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v = dpp_add<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Stage timing (include/mvi_raster.h: mvi_raster_timing_*). A StageTimer brackets the launches
// enqueued during its lifetime with two hipEvents when timing is enabled; otherwise it is free.
enum Stage { kStPreFwd = 0, kStScan, kStDup, kStSort, kStRanges, kStRenderFwd, kStRenderBwd, kStPreBwd };
void stage_begin(int stage, hipStream_t st);
void stage_end(int stage, hipStream_t st);
struct StageTimer {
    int stage; hipStream_t st;
    StageTimer(int s, hipStream_t q) : stage(s), st(q) { stage_begin(stage, st); }
    ~StageTimer() { stage_end(stage, st); }
};

// Memory a kernel zeroes on the side while it runs (the render kernels are issue-bound with the memory pipes idle): up to
// kMaxZero arrays of fp32 words, each cut into an unaligned head (< 4 words), a 16-byte-aligned body and a tail (< 4 words);
// the bodies are shared out over the blocks of the launch, block 0 also writes heads and tails.
constexpr int kMaxZero = 10;
struct ZeroRegions {
    float* head[kMaxZero];      // first word of the array
    uint4* body[kMaxZero];
    float* tail[kMaxZero];
    uint32_t n_head[kMaxZero], n_body[kMaxZero], n_tail[kMaxZero];   // words, 16-byte units, words
    int n = 0;
    bool overflow = false;      // an array was offered when all kMaxZero slots were taken: the caller must fail, not run
    void add(void* ptr, size_t words) {
        if (!ptr || !words) return;
        if (n >= kMaxZero) { overflow = true; return; }
        float* p = (float*)ptr;
        const size_t mis = ((uintptr_t)p & 15u) / 4u;
        size_t h = mis ? 4 - mis : 0;
        if (h > words) h = words;
        const size_t b = (words - h) / 4, t = words - h - 4 * b;
        head[n] = p; n_head[n] = (uint32_t)h;
        body[n] = (uint4*)(p + h); n_body[n] = (uint32_t)b;
        tail[n] = p + h + 4 * b; n_tail[n] = (uint32_t)t;
        ++n;
    }
};
__device__ __forceinline__ void zero_share(const ZeroRegions& z, int tid, int nthreads) {
    for (int r = 0; r < z.n; ++r) {
        const uint32_t per = (z.n_body[r] + gridDim.x - 1) / gridDim.x;
        const size_t z0 = (size_t)blockIdx.x * per;
        // write-through stores that do not stay in the XCD's L2 (MI355X_MICROARCH.md, stores of each flavour): the zeros are
        // not read again by this kernel, and the lines they would occupy hold the tile lists the kernel gathers from
        typedef uint32_t zero_u32x4 __attribute__((ext_vector_type(4)));
        const zero_u32x4 zz = {0u, 0u, 0u, 0u};
        for (uint32_t i = tid; i < per && z0 + i < z.n_body[r]; i += nthreads)
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(z.body[r] + z0 + i), "v"(zz) : "memory");
        if (blockIdx.x == 0) {
            if (tid < (int)z.n_head[r]) z.head[r][tid] = 0.f;
            if (tid < (int)z.n_tail[r]) z.tail[r][tid] = 0.f;
        }
    }
}
int launch_zero_regions(const ZeroRegions& z, hipStream_t st);

// launchers (defined in the .hip files; all enqueue on `st`, none synchronise)
int launch_preprocess_forward(const Frame& f, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities, const float* scales,
                              const float* rotations, const float* cov3D_precomp, GeomView g,
                              int32_t* radii, hipStream_t st);
// raw mode only: raw_opacity [P] (chain rule of the sigmoid), dL_dshs_rest [P,M-1,3]
struct RawBackwardExtra { const float* raw_opacity = nullptr; float* dL_dshs_rest = nullptr; int rows_prezeroed = 0; };
// sparse form: the gradient outputs are already zero (the render backward zeroed them on the side) and only the Gaussians
// whose accumulation row was touched are read, computed and written (per-lane accesses: ~3 % of the rows)
int launch_preprocess_backward_sparse(const Frame& f, const float* means3D, const float* shs, const float* scales,
                                      const float* rotations, const float* cov3D_precomp, GeomView g, const float* grad_rows,
                                      float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors,
                                      float* dL_dshs, float* dL_dcov3D, float* dL_dscales, float* dL_drots, hipStream_t st,
                                      RawBackwardExtra raw);
int launch_preprocess_backward(const Frame& f, const float* means3D, const float* shs,
                               const float* scales, const float* rotations, const float* cov3D_precomp,
                               const int32_t* radii, GeomView g, const float* grad_rows, float* dL_dmeans3D,
                               float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors, float* dL_dshs,
                               float* dL_dcov3D, float* dL_dscales, float* dL_drots, hipStream_t st,
                               RawBackwardExtra raw = RawBackwardExtra());
int launch_binning(const Frame& f, GeomView g, const int32_t* radii, BinningView b, ImageView im,
                   int64_t D, hipStream_t st);
int launch_zero_fill(void* p, size_t bytes, hipStream_t st);
int launch_binning2_totals(GeomView g, int P, unsigned long long* totals_host_devptr, hipStream_t st);   // [0] = D, [1] = segments
int launch_binning2_level1(const Frame& f, GeomView g, hipStream_t st);
// segments: the exact number of column segments if known (> 0), else the bound D is used to size the pass-2 grids
int launch_binning2(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D, int64_t segments, hipStream_t st);
int launch_render_forward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                          float* out_color, float* out_depth, hipStream_t st, float* zero_rows = nullptr);
int launch_render_backward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                           const float* dL_dpix, float* grad_rows, hipStream_t st, const ZeroRegions* zero = nullptr);
constexpr int kGradRow = 16;   // floats per Gaussian in the backward accumulation rows (64 B)
int launch_sh_backward_views(int P, int M, int deg, int n_views, const float* means3D, const float* campos,
                             int64_t campos_stride, const float* gcol, int64_t gcol_stride, float* dL_dshs, hipStream_t st);
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* visible, hipStream_t st);
int launch_color_factors(int P, const int32_t* radii, const float* grad_rows, const uint8_t* clamped, float* out, hipStream_t st);

}  // namespace mvi
